// planar_f32.hip -- the planar register kernels instantiated for float (one translation unit per precision: parallel builds)
#include "planar_impl.hpp"
namespace dartk {
std::unique_ptr<Impl> make_planar_impl_f32(const DartModelCard& c, std::string& why, bool allow_static) { return make_planar<float>(c, why, allow_static); }
}  // namespace dartk
