// the f32 instantiations of the register-resident mixed-radix kernels (mixed_radix_reg3.h: three stages, 300 ... 4096 points;
// mixed_radix_reg2.h: two stages, 20 ... 250 points)
#include "mixed_radix_reg3.h"
#include "mixed_radix_reg2.h"
namespace bdsp {
template int mr_reg3_launch<float>(const MrReg3Io<float>&, size_t, size_t, bool, hipStream_t);
template int mr_reg2_launch<float>(const MrReg3Io<float>&, size_t, size_t, bool, hipStream_t);
}
