// Toolchain probe: verifies a hipcc-7.2-built .so can launch on torch's HIP stream (ROCm 7.0 runtime)
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void axpy_k(const float* x, float* y, float a, int n){int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i]=a*x[i]+y[i];}
// one wave: C(32x32) = A(32xK) * B(KxN)^T-style with A[32][K], Bt[32][K] (both k-contiguous)
__global__ void mfma_k(const float* A, const float* Bt, float* C, int K){
  int l=threadIdx.x; int r=l&31, h=l>>5; f32x16 acc={0};
  for(int k=0;k<K;k+=2){ float a=A[r*K+k+h]; float b=Bt[r*K+k+h]; acc=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,acc,0,0,0);}
  for(int i=0;i<16;i++){int row=(i&3)+8*(i>>2)+4*h; C[row*32+r]=acc[i];}
}
extern "C" int probe_axpy(const float* x,float* y,float a,int n,void* stream){
  hipLaunchKernelGGL(axpy_k,dim3((n+255)/256),dim3(256),0,(hipStream_t)stream,x,y,a,n); return (int)hipGetLastError();}
extern "C" int probe_mfma(const float* A,const float* Bt,float* C,int K,void* stream){
  hipLaunchKernelGGL(mfma_k,dim3(1),dim3(64),0,(hipStream_t)stream,A,Bt,C,K); return (int)hipGetLastError();}
