// the f64 instantiations of the register-resident mixed-radix kernel (mixed_radix_reg3.h: three stages, 300 ... 2048 points).
// (The two-stage kernel for shorter lengths, mixed_radix_reg2.h, is f32 only: in f64 its two LDS buffers leave one workgroup per
// CU and it measured slower than k_mr_wg -- 65536 x 100 points 100 against 80 us.)
#include "mixed_radix_reg3.h"
namespace bdsp {
template int mr_reg3_launch<double>(const MrReg3Io<double>&, size_t, size_t, bool, hipStream_t);
}
