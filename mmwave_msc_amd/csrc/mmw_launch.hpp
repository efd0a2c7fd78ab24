// mmw_launch.hpp -- kernel launch with optional start/stop events carried by the dispatch itself.
//
// Timing a kernel with hipEventRecord before and after its launch puts two marker packets into the stream and
// costs ~10 us of idle time per pair.  hipExtLaunchKernel attaches the two events to the kernel's own AQL packet
// instead: same hipEventElapsedTime afterwards, no extra packets, so bench.py can time launches inside its
// timed region without stretching it.  mmw_api.hip arms `g_launch_prof` right before a launch_* call; the one
// launch that follows consumes it.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace mmw {

struct LaunchProf { hipEvent_t a = nullptr, b = nullptr; };
extern thread_local LaunchProf g_launch_prof;

#ifdef MMW_DIAG_POISON
// Diagnostic build (make DIAG=poison DIAGFLAGS=-DMMW_DIAG_POISON; never the product): every step kernel is preceded by a launch
// that fills the LDS of every CU with a NaN / huge-integer pattern, and mmw_create fills every device buffer with 0xFF before its
// own initialisation -- a read of LDS or global memory that nothing has written shows up in the parity tests instead of
// depending on what the previous occupant of the memory left behind (a foreign process's kernel, on a shared GPU).
__global__ void k_poison_lds();
void launch_poison(hipStream_t stream);
#endif

template <typename F, typename... Args>
inline void mmw_launch(F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args)
{
#ifdef MMW_DIAG_POISON
    launch_poison(stream);
#endif
    if (g_launch_prof.a) {
        const LaunchProf p = g_launch_prof;
        g_launch_prof = LaunchProf{};
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, stream, p.a, p.b, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    }
}

}  // namespace mmw
