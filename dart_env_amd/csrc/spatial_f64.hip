// spatial_f64.hip -- the tree kernel instantiated for double (one translation unit per precision: parallel builds)
#include "spatial_impl.hpp"
namespace dartk {
std::unique_ptr<Impl> make_spatial_impl_f64(const DartModelCard& c, std::string& why) { return make_spatial<double>(c, why); }
int dyn_prepare_f64(const DartModelCard& c, DynModel& out, std::string& err) { return dyn_prepare<double>(c, out, err); }
hipError_t dyn_launch_f64(hipStream_t s, const DynModel& m, int64_t n, const void* q, const void* dq, int soa, double* mass, double* bias,
                          double* pose, int nbodies) { return dyn_launch<double>(s, m, n, q, dq, soa, mass, bias, pose, nbodies); }
}  // namespace dartk
