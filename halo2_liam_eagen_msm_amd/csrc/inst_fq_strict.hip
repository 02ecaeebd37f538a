// Explicit instantiations of the heavy kernels for one (field, arithmetic) pair: its own
// translation unit so that the library builds in parallel.
#include "kernels_ec.cuh"
using namespace lemsm;
typedef XYZZ<Field32<FqParams>> G;
template __global__ void lemsm::k_accum1<G, 4>(GroupPlan, const u32*, const u32*, u32*, const uint4*, char*, u32*, char*);
template __global__ void lemsm::k_merge_pairs<G>(GroupPlan, u32, MqLayout, const u32*, const u32*, const u32*, const char*, char*, u32*, uint4*);
template __global__ void lemsm::k_merge_serial<G>(u32, MqLayout, const u32*, const uint4*, const char*, char*);
template __global__ void lemsm::k_merge_waves<G>(u32, MqLayout, const u32*, const uint4*, const char*, char*, char*, u32*);
template __global__ void lemsm::k_merge_final<G>(u32, MqLayout, const u32*, const uint4*, const char*, char*, u32*);
template __global__ void lemsm::k_pyramid<G>(const PyrTask*, u32, u32, u32, char*, u32);
template __global__ void lemsm::k_pyramid_first2<G>(PyrFirst2Args, const u32*, char*, u32*);
template __global__ void lemsm::k_pyramid_tail<G>(const PyrTask*, PyrTailArgs, const CopyTaskPod*, u32, char*, u32);
