// impl_iface.hpp -- the interface between the C ABI (dart_stepper.hip) and the per-precision kernel translation units.
// One heavy translation unit per (kernel family, precision) keeps a full rebuild parallel: planar_f32.hip, planar_f64.hip,
// spatial_f32.hip, spatial_f64.hip each instantiate their kernels and export one factory below.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dart_stepper.h"

namespace dartk {

struct Impl {
  virtual ~Impl() {}
  virtual hipError_t step(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const float* act,
                          float* obs, float* rew, uint8_t* done, uint8_t* trunc, int autoreset, uint64_t seed,
                          uint64_t off) = 0;
  virtual hipError_t reset(hipStream_t s, int64_t n, void* q, void* dq, int32_t* el, uint32_t* ep, const uint8_t* mask,
                           const double* qn, const double* vn, float* obs, uint64_t seed, uint64_t off,
                           int obs_masked_only = 0) = 0;
  virtual hipError_t state_io(hipStream_t s, int64_t n, void* q, void* dq, double* qh, double* dqh, int to_device) = 0;
  virtual void set_solver(int solver, int it1, int it2) = 0;
  virtual void set_stats(unsigned long long* p) = 0;
  virtual void set_force_slow(int /*on*/) {}   // planar kernels: route every touching env through the single-lane fallback solver (tests)
  // the MT19937 bank's device view (mt19937_draw.hpp: MtBankView*), or null.  true: this implementation's step kernel resets a finished env
  // from the bank itself when `autoreset` is on -- no mt_draw / reset launches behind it; false (default): it cannot, the caller runs those
  virtual bool set_mt_bank(const void* /*d_view*/) { return false; }
  virtual hipError_t prepare(int64_t) { return hipSuccess; }   // per-handle device allocations of the implementation
  virtual void release() {}
  virtual hipError_t debug_dump(double*) { return hipErrorInvalidValue; }
  virtual int set_launch_order(int /*longest_first*/) { return 0; }   // DART_CFG_LAUNCH_ORDER: tree kernel; the lane kernels have no use for it
  virtual int set_ext_force(int /*body*/, const double* /*host_force*/, int64_t /*n*/) { return DART_E_UNSUPPORTED; }
  virtual int set_task_state(hipStream_t, const uint8_t* /*d_mask*/, const double* /*d_values*/, int64_t /*n*/) { return DART_E_UNSUPPORTED; }
  virtual int slots() const = 0;
  virtual int max_contacts() const { return 0; }
  virtual int64_t lds_bytes() const { return 0; }
  // device buffers of the implementation that persist between steps (dart_snapshot / dart_restore)
  virtual void persistent(std::vector<std::pair<void*, size_t>>&, int64_t /*n*/) {}
  virtual int set_contact_report(bool /*on*/, int64_t /*n*/) { return DART_E_UNSUPPORTED; }
  virtual int get_contacts(hipStream_t, int64_t /*n*/, int32_t* /*count*/, int32_t* /*bodies*/, double* /*point_force*/, int /*max*/) { return DART_E_UNSUPPORTED; }
  virtual int get_constraint_forces(hipStream_t, int64_t /*n*/, double* /*out*/) { return DART_E_UNSUPPORTED; }
  bool soa = true;         // state layout: q[n][N] (planar kernels) or q[N][n] (spatial kernel)
  int block_threads = 64;  // active lanes per wave64 workgroup (32 -> twice the waves)
  bool lane_kernel = false;   // true: one env per GPU lane (planar / cart / arm / chain kernels); false: one env per wavefront (tree kernel)
  bool is_static = false;  // true: model constants are compile-time immediates (static_models.hpp) / the tree's factor pattern is (tree_patterns.hpp)
};

// factories of the kernel translation units: nullptr + a reason appended to `why` when the card does not fit
std::unique_ptr<Impl> make_planar_impl_f32(const DartModelCard& c, std::string& why, bool allow_static);
std::unique_ptr<Impl> make_planar_impl_f64(const DartModelCard& c, std::string& why, bool allow_static);
std::unique_ptr<Impl> make_spatial_impl_f32(const DartModelCard& c, std::string& why);
std::unique_ptr<Impl> make_spatial_impl_f64(const DartModelCard& c, std::string& why);

// dynamics getters (dart_get_dynamics / dart_get_body_poses): a physics-only device model of the card + one launch
struct DynModel { void* dev = nullptr; size_t lds = 0; bool free_root = false; };
int dyn_prepare_f32(const DartModelCard& c, DynModel& out, std::string& err);
int dyn_prepare_f64(const DartModelCard& c, DynModel& out, std::string& err);
hipError_t dyn_launch_f32(hipStream_t s, const DynModel& m, int64_t n, const void* q, const void* dq, int soa, double* mass,
                          double* bias, double* pose, int nbodies);
hipError_t dyn_launch_f64(hipStream_t s, const DynModel& m, int64_t n, const void* q, const void* dq, int soa, double* mass,
                          double* bias, double* pose, int nbodies);

}  // namespace dartk
