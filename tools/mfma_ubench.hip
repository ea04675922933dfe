// Micro-benchmark: int8 MFMA issue rate on gfx950 (32x32x32 and 16x16x64), 1..2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
template<int MODE>
__global__ __launch_bounds__(256) void k(int* out, int iters){
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
  v16i c0={0},c1={0},c2={0},c3={0};
  v4i d0={0},d1={0},d2={0},d3={0};
  for(int it=0; it<iters; ++it){
    if (MODE==0){
      c0=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c0,0,0,0);
      c1=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c1,0,0,0);
      c2=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c2,0,0,0);
      c3=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c3,0,0,0);
    } else if (MODE==1) { // single dependent chain
      c0=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c0,0,0,0);
      c0=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c0,0,0,0);
      c0=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c0,0,0,0);
      c0=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c0,0,0,0);
    } else {
      d0=__builtin_amdgcn_mfma_i32_16x16x64_i8(a,b,d0,0,0,0);
      d1=__builtin_amdgcn_mfma_i32_16x16x64_i8(a,b,d1,0,0,0);
      d2=__builtin_amdgcn_mfma_i32_16x16x64_i8(a,b,d2,0,0,0);
      d3=__builtin_amdgcn_mfma_i32_16x16x64_i8(a,b,d3,0,0,0);
    }
  }
  int s=0; for(int i=0;i<16;i++) s+=c0[i]+c1[i]+c2[i]+c3[i]; for(int i=0;i<4;i++) s+=d0[i]+d1[i]+d2[i]+d3[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int MODE> int run(int wg_per_cu, const char* name, double ops_per_mfma){
  int nb = 256*wg_per_cu; int iters=20000; int* d; CK(hipMalloc(&d,(nb*256)*4));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<nb,256>>>(d,iters); CK(hipDeviceSynchronize());
  hipEventRecord(a); k<MODE><<<nb,256>>>(d,iters); hipEventRecord(b); CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms,a,b);
  double mf = (double)iters*4;            // MFMAs per wave
  double per_simd = mf*wg_per_cu;         // MFMAs per SIMD
  printf("%-26s waves/SIMD=%d time=%.3f ms  ns per MFMA per SIMD=%.2f  chip rate=%.0f TOP/s\n", name, wg_per_cu, ms, ms*1e6/per_simd, per_simd*1024*ops_per_mfma/(ms*1e-3)/1e12);
  hipFree(d); return 0;
}
int main(){
  for(int w=1; w<=2; ++w){ run<0>(w,"i8 32x32x32 4 indep",65536.0); run<1>(w,"i8 32x32x32 1 chain",65536.0); run<2>(w,"i8 16x16x64 4 indep",32768.0);} 
  return 0;
}
