#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const f2* a, const f2* b, f2* out) {
  f2 x = a[threadIdx.x], y = b[threadIdx.x], r0, r1, r2, r3;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0]" : "=v"(r0) : "v"(x), "v"(y));
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(r1) : "v"(x), "v"(y));
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(r2) : "v"(x), "v"(y));
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0]" : "=v"(r3) : "v"(x), "v"(y));
  out[4*threadIdx.x]=r0; out[4*threadIdx.x+1]=r1; out[4*threadIdx.x+2]=r2; out[4*threadIdx.x+3]=r3;
}
int main(){
  f2 ha={1,2}, hb={3,4}, *a,*b,*o; f2 ho[4];
  hipMalloc(&a,8); hipMalloc(&b,8); hipMalloc(&o,32);
  hipMemcpy(a,&ha,8,hipMemcpyHostToDevice); hipMemcpy(b,&hb,8,hipMemcpyHostToDevice);
  k<<<1,1>>>(a,b,o); hipMemcpy(ho,o,32,hipMemcpyDeviceToHost);
  const char* names[4]={"sel[0,0] hi[0,0]","sel[1,1] hi[1,1]","sel[0,1] hi[0,1]","sel[1,0] hi[1,0]"};
  for(int i=0;i<4;i++) printf("%s -> {%g,%g}\n",names[i],ho[i].x,ho[i].y);
}
