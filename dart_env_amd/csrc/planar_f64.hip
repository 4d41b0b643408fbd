// planar_f64.hip -- the planar register kernels instantiated for double (one translation unit per precision: parallel builds)
#include "planar_impl.hpp"
namespace dartk {
std::unique_ptr<Impl> make_planar_impl_f64(const DartModelCard& c, std::string& why, bool allow_static) { return make_planar<double>(c, why, allow_static); }
}  // namespace dartk
