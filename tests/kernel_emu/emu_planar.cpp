// TEST INFRASTRUCTURE -- the planar register kernels compiled for the host (see fake_include/hip/hip_runtime.h) behind a
// minimal C interface: create / reset / step / state on host arrays laid out exactly like the device buffers.
// Built by tests/kernel_emu/Makefile into libdart_planar_emu.so; loaded only by tests/emu_lib.py.
#include <memory>
#include <string>
#include <vector>

#ifdef EMU_TREE_KERNEL   // libdart_spatial_emu.so: the same interface over the tree kernel, on the fiber runtime of fake_wave_include/
#include "spatial_impl.hpp"
using namespace dartk;
namespace dartk {
std::unique_ptr<Impl> make_planar_impl_f32(const DartModelCard& c, std::string& why, bool) { return make_spatial<float>(c, why); }
std::unique_ptr<Impl> make_planar_impl_f64(const DartModelCard& c, std::string& why, bool) { return make_spatial<double>(c, why); }
}
#else
#include "planar_impl.hpp"

using namespace dartk;

namespace dartk {   // the spatial factories live in TUs this library does not build
std::unique_ptr<Impl> make_planar_impl_f32(const DartModelCard& c, std::string& why, bool s) { return make_planar<float>(c, why, s); }
std::unique_ptr<Impl> make_planar_impl_f64(const DartModelCard& c, std::string& why, bool s) { return make_planar<double>(c, why, s); }
}
#endif

struct Emu {
  DartModelCard card;
  int64_t n; int precision;
  std::unique_ptr<Impl> impl;
  std::vector<unsigned char> q, dq;
  std::vector<int32_t> elapsed; std::vector<uint32_t> episode;
  std::vector<double> qn, vn;
  std::vector<unsigned long long> stats;
  std::string err;
};

extern "C" {
const char* emu_last_error(Emu* h) { static std::string g; return h ? h->err.c_str() : g.c_str(); }
Emu* emu_create(const DartModelCard* card, int64_t n, int precision, int allow_static, char* why_out, int why_len) {
  std::string why;
  auto impl = precision == 32 ? make_planar_impl_f32(*card, why, allow_static != 0) : make_planar_impl_f64(*card, why, allow_static != 0);
  if (!impl) { if (why_out) snprintf(why_out, why_len, "%s", why.c_str()); return nullptr; }
  auto* h = new Emu();
  h->card = *card; h->n = n; h->precision = precision; h->impl = std::move(impl);
  h->impl->set_solver(0, 0, 0);
  (void)h->impl->prepare(n);
  const size_t rs = precision == 32 ? 4 : 8, nd = card->ndofs;
  h->q.assign(rs * nd * n, 0); h->dq.assign(rs * nd * n, 0);
  h->elapsed.assign(n, 0); h->episode.assign(n, 0);
  h->qn.resize(nd * n); h->vn.resize(nd * n);
  for (int64_t i = 0; i < n; i++) for (size_t d = 0; d < nd; d++) { h->qn[i * nd + d] = card->init_pos[d]; h->vn[i * nd + d] = card->init_vel[d]; }
  h->impl->state_io(nullptr, n, h->q.data(), h->dq.data(), h->qn.data(), h->vn.data(), 1);
  return h;
}
void emu_destroy(Emu* h) { if (h) { h->impl->release(); delete h; } }
int emu_is_static(Emu* h) { return h->impl->is_static ? 1 : 0; }
int emu_slots(Emu* h) { return h->impl->slots(); }
#ifdef EMU_TREE_KERNEL
// the tree kernel's LDS block: bytes the host reserves (sp_lds_bytes) and the end of what the kernels carve out of it (sp_carve) for the same
// dimensions -- tests/test_tree_kernel_emu_parity.py sweeps them against each other (a carve that runs past the reservation is silent
// corruption of the neighbouring workgroup's block on the device)
// hreals: Reals of the H block (0 = the padded dense rows; a pattern kernel's skyline size otherwise -- SpatialModel::hreals)
long long emu_lds_reserved(int nl, int n, int real_bytes, int maxm, int maxcp, int reg_lcp, int hreals) { return (long long)sp_lds_bytes(nl, n, (size_t)real_bytes, maxm, maxcp, reg_lcp, hreals); }
long long emu_lds_carved(int nl, int n, int real_bytes, int maxm, int maxcp, int reg_lcp, int hreals) {
  alignas(16) static unsigned char base[1];
  if (real_bytes == 4) { auto S = sp_carve<float>((float*)base, nl, n, maxm, maxcp, reg_lcp, hreals); return (long long)((unsigned char*)(S.ticks + 10) - base); }
  auto S = sp_carve<double>((double*)base, nl, n, maxm, maxcp, reg_lcp, hreals); return (long long)((unsigned char*)(S.ticks + 10) - base);
}
int emu_pattern_hreals() { return HumanWalkerPattern::hreals; }
#endif
#ifdef DART_WAVE_EMU
// a static __shared__ array of this library (address = load base + symbol value, found by the test in the symbol table): poisoned with the
// dynamic block before every workgroup when DART_EMU_POISON_LDS is set
int emu_register_static_lds(void* addr, unsigned long long bytes) {
  if (wave_emu::n_static_lds >= 256) return -1;
  wave_emu::static_lds[wave_emu::n_static_lds] = (unsigned char*)addr; wave_emu::static_lds_bytes[wave_emu::n_static_lds++] = (size_t)bytes;
  return wave_emu::n_static_lds;
}
#endif
long long emu_lds_bytes(Emu* h) { return (long long)h->impl->lds_bytes(); }   // DART_Q_LDS_BYTES of the product library: the step kernel's LDS block
void emu_set_solver(Emu* h, int solver, int it1, int it2) { h->impl->set_solver(solver, it1, it2); }
int emu_set_ext_force(Emu* h, int body, const double* force) { return h->impl->set_ext_force(body, force, h->n); }
int emu_contact_report(Emu* h, int on) { return h->impl->set_contact_report(on != 0, h->n); }
int emu_max_contacts(Emu* h) { return h->impl->max_contacts(); }
int emu_get_contacts(Emu* h, int32_t* count, int32_t* bodies, double* point_force, int maxc) { return h->impl->get_contacts(nullptr, h->n, count, bodies, point_force, maxc); }
int emu_get_constraint_forces(Emu* h, double* out) { return h->impl->get_constraint_forces(nullptr, h->n, out); }
void emu_force_slow(Emu* h, int on) { h->impl->set_force_slow(on); }
int emu_set_task_state(Emu* h, const uint8_t* mask, const double* values4) { return h->impl->set_task_state(nullptr, mask, values4, h->n); }
void emu_enable_stats(Emu* h, int on) { h->stats.assign(64, 0); h->impl->set_stats(on ? h->stats.data() : nullptr); }
void emu_get_stats(Emu* h, unsigned long long* out64) { memcpy(out64, h->stats.data(), 64 * sizeof(unsigned long long)); }
// mask: n bytes or NULL; noise rows (n, ndofs) doubles or NULL (Philox with `seed`, `env_offset`)
void emu_reset(Emu* h, const uint8_t* mask, const double* qnoise, const double* vnoise, float* obs, uint64_t seed, uint64_t off, int obs_masked_only) {
  const size_t nd = h->card.ndofs;
  const double *qn = nullptr, *vn = nullptr;
  if (qnoise) {
    for (int64_t i = 0; i < h->n; i++) {
      if (mask && !mask[i]) continue;
      for (size_t d = 0; d < nd; d++) { h->qn[i * nd + d] = h->card.init_pos[d] + qnoise[i * nd + d]; h->vn[i * nd + d] = h->card.init_vel[d] + vnoise[i * nd + d]; }
    }
    qn = h->qn.data(); vn = h->vn.data();
  }
  h->impl->reset(nullptr, h->n, h->q.data(), h->dq.data(), h->elapsed.data(), h->episode.data(), mask, qn, vn, obs, seed, off, obs_masked_only);
}
void emu_step(Emu* h, const float* actions, float* obs, float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed, uint64_t off) {
  h->impl->step(nullptr, h->n, h->q.data(), h->dq.data(), h->elapsed.data(), h->episode.data(), actions, obs, rew, done, trunc, autoreset, seed, off);
}
void emu_state(Emu* h, double* q, double* dq, int to_device) { h->impl->state_io(nullptr, h->n, h->q.data(), h->dq.data(), q, dq, to_device); }
void emu_counters(Emu* h, int32_t* el, uint32_t* ep) { memcpy(el, h->elapsed.data(), 4 * h->n); memcpy(ep, h->episode.data(), 4 * h->n); }
}
