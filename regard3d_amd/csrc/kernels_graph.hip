// kernels_graph.hip -- the device mirror of a match graph (GraphDev, r3dm_ctx.hpp): segments of matches gathered on the device.
//
// A match graph is produced on the device twice over -- the finalisation kernel appends every pair's matches to one array in whatever
// order the workgroups finish, the filters leave per-pair lists of inlier INDICES into the putative matches -- and is copied to the
// host for the files and the SfM stage that follow (PairWiseMatches is a std::map in the reference,
// /root/reference/src/R3DComputeMatches.cpp:481-487).  A host with one process per GPU also needs it on the wire
// (r3dm_allgather_graphs): these kernels lay the kept pairs' matches out in graph order in device memory, so the exchange sends
// device buffers and nothing of the payload crosses PCIe on the way out.
#include "r3dm_internal.hpp"

namespace r3dm {

// dst[seg.dst + q] = src[seg.src + (idx ? idx[seg.idx + q] : q)], q < seg.cnt; one workgroup per segment (grid-stride over segments)
__global__ __launch_bounds__(256)
void graph_gather_kernel(const r3dm_match* __restrict__ src, const uint32_t* __restrict__ idx, const GraphSeg* __restrict__ segs, uint32_t n_segs,
                         r3dm_match* __restrict__ dst)
{
    for (uint32_t s = blockIdx.x; s < n_segs; s += gridDim.x) {
        const GraphSeg g = segs[s];
        const r3dm_match* __restrict__ from = src + g.src;
        r3dm_match* __restrict__ to = dst + g.dst;
        if (idx) {
            const uint32_t* __restrict__ ix = idx + g.idx;
            for (uint32_t q = threadIdx.x; q < g.cnt; q += 256u) to[q] = from[ix[q]];
        } else {
            for (uint32_t q = threadIdx.x; q < g.cnt; q += 256u) to[q] = from[q];
        }
    }
}

hipError_t launch_graph_gather(hipStream_t st, const r3dm_match* src, const uint32_t* idx, const GraphSeg* segs, uint32_t n_segs, r3dm_match* dst)
{
    if (n_segs == 0) return hipSuccess;
    hipLaunchKernelGGL(graph_gather_kernel, dim3(n_segs < 4096u ? n_segs : 4096u), dim3(256), 0, st, src, idx, segs, n_segs, dst);
    return hipGetLastError();
}

}  // namespace r3dm
