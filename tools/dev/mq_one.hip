// developer scratch: ONE instantiation of ls_mq_kernel for ISA inspection
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -DMQ_L=64 -DMQ_V=4 -DMQ_M=3 -DMQ_NB=2 --save-temps -c tools/dev/mq_one.hip
#define LS_MQ_KERNEL_ONLY
#include "../../lean-explore_amd/csrc/ls_mq.hip"
template __global__ void ls_mq_kernel<MQ_L, MQ_V, MQ_M, MQ_NB, MQ_WPB>(
    const mq_f32x4*, long long, const float*, int, int, int, float*, long long, u64*, long long, u64*, long long, int, int,
    ls_fin_batch, void*, long long, u32, float*);
