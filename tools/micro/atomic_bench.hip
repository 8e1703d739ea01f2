// Microbenchmark: fp32 atomic-add throughput patterns on MI355X (development aid for the hash backward).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__device__ __forceinline__ uint32_t hash32(uint32_t x){ x ^= x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

// mode 0: random 4B per lane over `span` floats
// mode 1: lane pairs -> adjacent floats (8B), random pair location
// mode 2: 64 lanes contiguous (256B) at random 256B-aligned location
// mode 3: random 8B double atomics
// mode 4: random 4B, but 8 lanes share a random 32B sector (lane&7 offsets)
// mode 5: like 0 but with return value used
__global__ void k(float* buf, uint64_t span, int iters, int mode, float* sink){
    const uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x;
    const int lane = threadIdx.x & 63;
    float acc=0;
    for(int it=0; it<iters; ++it){
        uint32_t key = hash32(tid*2654435761u + it*40503u);
        uint32_t wkey = hash32((tid>>6)*2654435761u + it*40503u + 17u);
        if(mode==0){ atomicAdd(buf + (key % span), 1.0f); }
        else if(mode==1){ uint32_t pk = hash32((tid>>1)*2654435761u + it*40503u); atomicAdd(buf + ((uint64_t)(pk % (span/2))*2 + (lane&1)), 1.0f); }
        else if(mode==2){ atomicAdd(buf + ((uint64_t)(wkey % (span/64))*64 + lane), 1.0f); }
        else if(mode==3){ atomicAdd(reinterpret_cast<double*>(buf) + (key % (span/2)), 1.0); }
        else if(mode==4){ uint32_t sk = hash32((tid>>3)*2654435761u + it*40503u); atomicAdd(buf + ((uint64_t)(sk % (span/8))*8 + (lane&7)), 1.0f); }
        else if(mode==5){ acc += atomicAdd(buf + (key % span), 1.0f); }
        else if(mode==8){ // pair at lane stride 8: lanes l and l^8 hit adjacent floats
            uint32_t base = (tid & ~8u); uint32_t pk = hash32(base*2654435761u + it*40503u);
            atomicAdd(buf + ((uint64_t)(pk % (span/2))*2 + ((lane>>3)&1)), 1.0f); }
        else if(mode==9){ // quad at lane stride 8 within 32 lanes: lanes l, l^8, l^16, l^24 -> 4 adjacent floats (16B)
            uint32_t base = (tid & ~24u); uint32_t pk = hash32(base*2654435761u + it*40503u);
            atomicAdd(buf + ((uint64_t)(pk % (span/4))*4 + ((lane>>3)&3)), 1.0f); }
        else if(mode==10){ // half of the lanes inactive (exec-masked) random 4B
            if (lane & 1) atomicAdd(buf + (key % span), 1.0f); }
        else if(mode==6){ // plain scattered 4B store (no atomic) for reference
            buf[key % span] = 1.0f; }
        else if(mode==7){ // scattered 4B load
            acc += buf[key % span]; }
    }
    if(acc==123.f) sink[0]=acc;
}

int main(){
    const uint64_t big = 302ull*1000*1000;   // ~1.2 GB of floats
    float* buf; CK(hipMalloc(&buf, big*4)); CK(hipMemset(buf,0,big*4));
    float* sink; CK(hipMalloc(&sink,4));
    hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks=256*8, threads=256, iters=128;   // 67M lane-ops
    const double nops = (double)blocks*threads*iters;
    const char* names[]={"rand4B","pair8B","wave256B","rand8B_f64","sector32B","rand4B_ret","store4B","load4B","pair_s8","quad_s8","half_masked"};
    uint64_t spans[]={big};
    for(int mode=0; mode<11; ++mode){ if(mode>=2 && mode<=7) continue;
        for(uint64_t span: spans){
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, buf, span, 4, mode, sink); // warm
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, buf, span, iters, mode, sink);
            hipEventRecord(b); CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms,a,b);
            printf("%-12s span=%8.1f MB  %7.3f ms  %8.2f G lane-ops/s\n", names[mode], span*4/1e6, ms, nops/ms/1e6);
        }
    }
    return 0;
}
