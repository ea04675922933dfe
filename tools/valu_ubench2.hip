// Micro-benchmark: issue cost of integer vs float min / med3 on gfx950 (wave64), 1..4 waves per SIMD.
// Positive floats order like their bit patterns, so the matcher's unsigned top-2 keys could use either.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define OP2(name, d, a, b) asm volatile(name " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define OP3(name, d, a, b, c) asm volatile(name " %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
template<int MODE>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed){
  unsigned v[16], m1[16], m2[16];
  #pragma unroll
  for(int i=0;i<16;i++){ v[i]=(threadIdx.x*seed+i)&0x3fffff; m1[i]=0x7f000000u; m2[i]=0x7f000000u; }
  long long t0 = clock64();
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int i=0;i<16;i++){
      unsigned key = v[i] + it;
      if (MODE==0){ OP3("v_med3_u32", m2[i], m1[i], m2[i], key); OP2("v_min_u32", m1[i], m1[i], key); }
      if (MODE==1){ OP3("v_med3_f32", m2[i], m1[i], m2[i], key); OP2("v_min_f32", m1[i], m1[i], key); }
      if (MODE==2){ OP3("v_med3_i32", m2[i], m1[i], m2[i], key); OP2("v_min_i32", m1[i], m1[i], key); }
      if (MODE==3){ OP3("v_min3_u32", m2[i], m1[i], m2[i], key); OP3("v_min3_u32", m1[i], m1[i], key, m2[i]); }
      if (MODE==4){ OP3("v_min3_f32", m2[i], m1[i], m2[i], key); OP3("v_min3_f32", m1[i], m1[i], key, m2[i]); }
      if (MODE==5){ OP2("v_pk_min_u16", m2[i], m2[i], key); OP2("v_pk_min_u16", m1[i], m1[i], key); }
    }
  }
  long long t1 = clock64();
  unsigned s=0;
  #pragma unroll
  for(int i=0;i<16;i++) s+=m1[i]^m2[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  if(threadIdx.x==0) out[gridDim.x*blockDim.x + blockIdx.x] = (int)(t1-t0);
}
template<int MODE> int run(int wg_per_cu, const char* name){
  int nb = 256*wg_per_cu; int iters=20000; int* d; CK(hipMalloc(&d,(nb*256+nb)*4));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<nb,256>>>(d,iters,3); CK(hipDeviceSynchronize());
  hipEventRecord(a); k<MODE><<<nb,256>>>(d,iters,3); hipEventRecord(b); CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms,a,b);
  double ops = (double)iters*16*2 + (double)iters*16;  // two measured ops + the add
  printf("%-26s waves/SIMD=%d  time=%.3f ms  ns per wave-op per SIMD=%.3f\n", name, wg_per_cu, ms, ms*1e6/(ops*wg_per_cu));
  hipFree(d); return 0;
}
int main(){
  for(int w=1; w<=4; w*=2){
    run<0>(w,"med3_u32 + min_u32"); run<1>(w,"med3_f32 + min_f32"); run<2>(w,"med3_i32 + min_i32");
    run<3>(w,"min3_u32 x2"); run<4>(w,"min3_f32 x2"); run<5>(w,"pk_min_u16 x2"); }
  return 0;
}
