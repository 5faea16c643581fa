// the f64 instantiations of the register-resident mixed-radix kernel (mixed_radix_reg3.h)
#include "mixed_radix_reg3.h"
namespace bdsp { template int mr_reg3_launch<double>(const MrReg3Io<double>&, size_t, size_t, bool, hipStream_t); }
