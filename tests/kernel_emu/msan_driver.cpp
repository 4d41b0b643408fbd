// TEST INFRASTRUCTURE (not built by default) -- MemorySanitizer over the lane kernels: the host build of the device source (one lane after
// the other, fake_include/) as a stand-alone program on the scenario of tests/test_gpu_repeatability.py, every output checked for
// initialisedness.  Written in round 4 while looking for the cause of the half cheetah's first-launch difference on gfx950:
// clean -- no uninitialised value reaches a branch, an address or an output in the lane-at-a-time build.
//   python - <<'PY'      # inputs: the model card and the scenario's states / actions
//   import numpy as np; from dart_env_amd.model_card import card_for
//   c = card_for("DartHalfCheetah-v1"); n = 256; r = np.random.RandomState(5); open("/tmp/msan/card.bin", "wb").write(bytes(c))
//   q = r.uniform(-.3, .3, (n, c.ndofs)); dq = r.uniform(-2, 2, (n, c.ndofs)); q[:, 1] = r.uniform(-.65, -.3, n)
//   q.tofile("/tmp/msan/q0.bin"); dq.tofile("/tmp/msan/dq0.bin"); r.uniform(-1, 1, (5, n, c.act_dim)).astype(np.float32).tofile("/tmp/msan/acts.bin")
//   PY
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=memory -fsanitize-recover=memory -fsanitize-memory-track-origins=2 -std=c++17 -ffp-contract=fast \
//       -Ifake_include -I../../dart_env_amd/csrc -I../../include -I. msan_driver.cpp -o /tmp/msan/driver        (12 minutes)
//   MSAN_OPTIONS=halt_on_error=0 /tmp/msan/driver 32      (one false positive from the uninstrumented libstdc++ in emu_create is expected)
#include "emu_planar.cpp"
#include <cstdio>
#include <sanitizer/msan_interface.h>
static std::vector<unsigned char> slurp(const char* p) { FILE* f = fopen(p, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<unsigned char> b(n); fread(b.data(), 1, n, f); fclose(f); return b; }
int main(int argc, char** argv) {
  const int prec = argc > 1 ? atoi(argv[1]) : 32;
  auto cb = slurp("/tmp/msan/card.bin");
  DartModelCard card; memcpy(&card, cb.data(), sizeof(card));
  const int64_t n = 256; const int nd = card.ndofs, na = card.act_dim, no = card.obs_dim;
  auto q0 = slurp("/tmp/msan/q0.bin"), dq0 = slurp("/tmp/msan/dq0.bin"), ac = slurp("/tmp/msan/acts.bin");
  char why[256] = {0};
  Emu* h = emu_create(&card, n, prec, 1, why, 256);
  if (!h) { printf("create failed: %s\n", why); return 1; }
  emu_state(h, (double*)q0.data(), (double*)dq0.data(), 1);
  std::vector<float> obs(n * no), rew(n); std::vector<uint8_t> done(n), trunc(n);
  for (int t = 0; t < 5; t++) {
    emu_step(h, (const float*)ac.data() + (size_t)t * n * na, obs.data(), rew.data(), done.data(), trunc.data(), 0, 0, 0);
    __msan_check_mem_is_initialized(obs.data(), obs.size() * 4);
    __msan_check_mem_is_initialized(rew.data(), rew.size() * 4);
    __msan_check_mem_is_initialized(done.data(), done.size());
  }
  std::vector<double> q(n * nd), dq(n * nd);
  emu_state(h, q.data(), dq.data(), 0);
  __msan_check_mem_is_initialized(q.data(), q.size() * 8);
  __msan_check_mem_is_initialized(dq.data(), dq.size() * 8);
  double s = 0; for (double v : q) s += v; for (double v : dq) s += v;
  printf("done, checksum %.15g\n", s);
  return 0;
}
