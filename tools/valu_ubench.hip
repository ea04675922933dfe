// Micro-benchmark: VALU issue cost of the epilogue instruction mix on gfx950, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__device__ __forceinline__ int med3(int a,int b,int c){ return max(min(a,b), min(max(a,b),c)); }

template<int MODE>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed){
  int v[16], m1[16], m2[16];
  #pragma unroll
  for(int i=0;i<16;i++){ v[i]=threadIdx.x*seed+i; m1[i]=0x7fffffff; m2[i]=0x7fffffff; }
  long long t0 = clock64();
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int i=0;i<16;i++){
      if (MODE==0){ m1[i] = min(m1[i], v[i]+it); }                     // add+min : 2 ops
      if (MODE==1){ int k=(v[i]<<8)+it; m2[i]=med3(m1[i],m2[i],k); m1[i]=min(m1[i],k);} // lshl_add, med3, min : 3 ops
      if (MODE==2){ m1[i] = m1[i]*3 + v[i]; }                           // mad
    }
  }
  long long t1 = clock64();
  int s=0;
  #pragma unroll
  for(int i=0;i<16;i++) s+=m1[i]^m2[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  if(threadIdx.x==0) out[gridDim.x*blockDim.x + blockIdx.x] = (int)(t1-t0);
}
template<int MODE> int run(int wg_per_cu, const char* name, int ops_per_iter){
  int nb = 256*wg_per_cu; int iters=20000; int* d; CK(hipMalloc(&d,(nb*256+nb)*4));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<nb,256>>>(d,iters,3); CK(hipDeviceSynchronize());
  hipEventRecord(a); k<MODE><<<nb,256>>>(d,iters,3); hipEventRecord(b); CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms,a,b);
  std::vector<int> h(nb); CK(hipMemcpy(h.data(), d+nb*256, nb*4, hipMemcpyDeviceToHost));
  double cyc=0; for(int x: h) cyc+=x; cyc/=nb;
  // waves per SIMD = wg_per_cu (each WG = 4 waves = 1 per SIMD)
  double ops = (double)iters*16*ops_per_iter;
  printf("%-22s waves/SIMD=%d  time=%.3f ms  clock64 ticks/wave-op=%.2f   ns per wave-op per SIMD=%.3f\n", name, wg_per_cu, ms, cyc/ops, ms*1e6/(ops*wg_per_cu));
  hipFree(d); return 0;
}
int main(){
  for(int w=1; w<=4; w*=2){ run<0>(w,"add+min (2 ops)",2); run<1>(w,"lshl_add+med3+min (3)",3); run<2>(w,"mad (1 op)",1);}
  return 0;
}
