#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void k_stream(const uint4* __restrict__ a, uint64_t n4, uint32_t* out){
  uint64_t i = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x; uint64_t st=(uint64_t)gridDim.x*blockDim.x; uint32_t s=0;
  for(; i<n4; i+=st){ uint4 v=a[i]; s+=v.x^v.y^v.z^v.w; }
  if(s==0x12345678) out[0]=s;
}
// wave per read, 1 KiB chunks aligned to 1 KiB, no compute; PF = chunks in flight
template<int PF>
__global__ void k_read(const uint32_t* __restrict__ cig, const uint64_t* __restrict__ off, uint64_t n_reads, uint32_t* out){
  int lane=threadIdx.x&63; uint64_t w=(uint64_t)blockIdx.x*(blockDim.x>>6)+(threadIdx.x>>6); uint64_t st=(uint64_t)gridDim.x*(blockDim.x>>6); uint32_t s=0;
  for(uint64_t r=w;r<n_reads;r+=st){
    uint64_t c0=off[r], c1=off[r+1]; uint64_t base=c0&~255ull;
    for(uint64_t ch=base; ch<c1; ch+=256*PF){
      uint4 v[PF];
      #pragma unroll
      for(int p=0;p<PF;p++){ uint64_t idx=ch+256*p+lane*4; if(idx<c1) v[p]=*(const uint4*)(cig+idx); else v[p]=make_uint4(0,0,0,0);} 
      #pragma unroll
      for(int p=0;p<PF;p++) s+=v[p].x^v[p].y^v[p].z^v[p].w;
    }
  }
  if(s==0x12345678) out[0]=s;
}
// block per contiguous slab: flat
__global__ void k_slab(const uint4* __restrict__ a, uint64_t n4, uint64_t slab4, uint32_t* out){
  uint64_t b0=(uint64_t)blockIdx.x*slab4; uint32_t s=0;
  for(uint64_t i=b0+threadIdx.x;i<b0+slab4 && i<n4;i+=blockDim.x){uint4 v=a[i]; s+=v.x^v.y^v.z^v.w;}
  if(s==0x12345678) out[0]=s;
}
int main(){
  const uint64_t n_reads=129416; std::mt19937_64 g(1); std::vector<uint64_t> off(n_reads+1); off[0]=0;
  for(uint64_t r=0;r<n_reads;r++){ double L=exp(9.2+0.6*std::normal_distribution<double>(0,1)(g)); if(L<1000)L=1000; if(L>200000)L=200000; off[r+1]=off[r]+(uint64_t)(L*0.0842); }
  uint64_t m=off[n_reads]; printf("words %lu (%.1f MB)\n",(unsigned long)m,m*4/1e6);
  uint32_t* d; uint64_t* doff; uint32_t* dout; CK(hipMalloc(&d,m*4+4096)); CK(hipMalloc(&doff,(n_reads+1)*8)); CK(hipMalloc(&dout,64));
  CK(hipMemset(d,1,m*4+4096)); CK(hipMemcpy(doff,off.data(),(n_reads+1)*8,hipMemcpyHostToDevice));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); float ms;
  auto T=[&](const char* name, auto f){ f(); hipDeviceSynchronize(); hipEventRecord(a); for(int i=0;i<10;i++) f(); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("%-28s %.3f ms  %.2f TB/s\n",name,ms/10,m*4/(ms/10*1e-3)/1e12); };
  T("stream grid2048x256",[&]{ k_stream<<<2048,256>>>((const uint4*)d,m/4,dout);});
  T("stream grid8192x256",[&]{ k_stream<<<8192,256>>>((const uint4*)d,m/4,dout);});
  T("slab 64KB/block 256thr",[&]{ uint64_t slab4=4096; k_slab<<<(unsigned)((m/4+slab4-1)/slab4),256>>>((const uint4*)d,m/4,slab4,dout);});
  T("slab 16KB/block 64thr",[&]{ uint64_t slab4=1024; k_slab<<<(unsigned)((m/4+slab4-1)/slab4),64>>>((const uint4*)d,m/4,slab4,dout);});
  T("read PF1 1536x256",[&]{ k_read<1><<<1536,256>>>(d,doff,n_reads,dout);});
  T("read PF1 2048x256",[&]{ k_read<1><<<2048,256>>>(d,doff,n_reads,dout);});
  T("read PF2 2048x256",[&]{ k_read<2><<<2048,256>>>(d,doff,n_reads,dout);});
  T("read PF4 2048x256",[&]{ k_read<4><<<2048,256>>>(d,doff,n_reads,dout);});
  T("read PF4 4096x256",[&]{ k_read<4><<<4096,256>>>(d,doff,n_reads,dout);});
  T("read PF1 8192x256 (1 read/wave)",[&]{ k_read<1><<<8192,256>>>(d,doff,n_reads,dout);});
  T("read PF4 32354x256 (1 read/wave)",[&]{ k_read<4><<<32354,256>>>(d,doff,n_reads,dout);});
  return 0;
}
