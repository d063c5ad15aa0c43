// TEST INFRASTRUCTURE: storage for the SIMT emulation shim (see hip_emu.h).
#include "hip_emu.h"
namespace emu {
thread_local dim3 t_threadIdx;
thread_local dim3 t_blockIdx;
thread_local BlockCtx* t_block = nullptr;
thread_local int t_tid = 0;
dim3 g_blockDim, g_gridDim;
}  // namespace emu
