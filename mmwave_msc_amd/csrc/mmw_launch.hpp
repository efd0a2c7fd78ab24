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

template <typename F, typename... Args>
inline void mmw_launch(F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args)
{
    if (g_launch_prof.a) {
        const LaunchProf p = g_launch_prof;
        g_launch_prof = LaunchProf{};
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, stream, p.a, p.b, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    }
}

}  // namespace mmw
