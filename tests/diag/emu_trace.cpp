// Diagnostic build of the planar-kernel host emulation that records, for every run of the pivoting LCP loop, the active sets it
// started from and ended with (research into how often -- and why -- a lane needs more than one factorisation).
//   g++ -O2 -fPIC -std=c++17 -march=native -ffp-contract=fast -Wno-unused-value -Wno-attributes -Itests/kernel_emu/fake_include \
//       -Idart_env_amd/csrc -shared -o /tmp/libdart_planar_emu_trace.so tests/diag/emu_trace.cpp
//   DART_EMU_LIB=/tmp/libdart_planar_emu_trace.so python tests/diag/diag_lcp_active_sets.py DartWalker2d-v1
#include <stdint.h>
#include <stddef.h>
#include <vector>
#define DART_EMU_TRACE 1
struct TraceRec { uint32_t M, zero_bounds, F0, U0, F1, U1, pin, iters, env, substep; };
static std::vector<TraceRec> g_trace;
static int g_env = 0, g_substep = 0;
static inline void dart_emu_substep(int env, int f) { g_env = env; g_substep = f; }
static inline void dart_emu_trace(int M, int zb, uint32_t F0, uint32_t U0, uint32_t F1, uint32_t U1, uint32_t pin, int it) {
  if (g_trace.size() < (size_t)8000000) g_trace.push_back({(uint32_t)M, (uint32_t)zb, F0, U0, F1, U1, pin, (uint32_t)it, (uint32_t)g_env, (uint32_t)g_substep});
}
#include "../kernel_emu/emu_planar.cpp"
extern "C" {
int64_t emu_trace_size() { return (int64_t)g_trace.size(); }
void emu_trace_get(uint32_t* out) { for (size_t i = 0; i < g_trace.size(); i++) { const TraceRec& r = g_trace[i]; uint32_t* o = out + 8 * i; o[0] = r.M; o[1] = r.zero_bounds; o[2] = r.F0; o[3] = r.U0; o[4] = r.F1; o[5] = r.U1; o[6] = r.pin; o[7] = r.iters; } }
void emu_trace_get10(uint32_t* out) { for (size_t i = 0; i < g_trace.size(); i++) { const TraceRec& r = g_trace[i]; uint32_t* o = out + 10 * i; o[0] = r.M; o[1] = r.zero_bounds; o[2] = r.F0; o[3] = r.U0; o[4] = r.F1; o[5] = r.U1; o[6] = r.pin; o[7] = r.iters; o[8] = r.env; o[9] = r.substep; } }
void emu_trace_clear() { g_trace.clear(); }
}
