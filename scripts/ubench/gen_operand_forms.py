#!/usr/bin/env python3
"""Generates operand_forms.hip: VALU issue cost on gfx950 as a function of the operand form
(how many VGPR source dwords an instruction reads, packed vs scalar f32, in-place vs 3-address,
VGPR bank placement).  Every test is one asm block of 64 independent instructions on fixed
physical registers v0..v95, repeated ITERS times; cycles per wave-instruction per SIMD are
derived from wall time at W waves per SIMD.

    python gen_operand_forms.py > operand_forms.hip
    hipcc --offload-arch=gfx950 -O3 -o operand_forms operand_forms.hip
"""
NI = 64

def block(fmt):
    return [fmt(i) for i in range(NI)]

tests = {}
# 1 VGPR source dword, in place
tests["v_mul_f32 d,s,d          (1 vsrc)"] = block(lambda i: f"v_mul_f32 v{i}, s20, v{i}")
# VOP2 fmac: reads src1 and dst
tests["v_fmac_f32 d,s,b         (2 vsrc)"] = block(lambda i: f"v_fmac_f32 v{i}, s20, v{(i + 17) % 64}")
# VOP3 fma, SGPR + 2 VGPR (3-address, dst = one source)
tests["v_fma_f32 d,s,b,d        (2 vsrc)"] = block(lambda i: f"v_fma_f32 v{i}, s20, v{(i + 17) % 64}, v{i}")
tests["v_fma_f32 d,-s,b,d neg   (2 vsrc)"] = block(lambda i: f"v_fma_f32 v{i}, -s20, v{(i + 17) % 64}, v{i}")
# VOP3 fma out of place into a third register
tests["v_fma_f32 t,s,b,c        (2 vsrc)"] = block(lambda i: f"v_fma_f32 v{64 + (i % 32)}, s20, v{(i + 17) % 64}, v{i}")
# three VGPR sources, different banks (i, i+1, i+2)
tests["v_fma_f32 d,a,b,d 3vgpr banks 0,1,2"] = block(lambda i: f"v_fma_f32 v{i}, v{(i + 1) % 64}, v{(i + 2) % 64}, v{i}")
# three VGPR sources, same bank (i, i+4, i+8)
tests["v_fma_f32 d,a,b,d 3vgpr same bank  "] = block(lambda i: f"v_fma_f32 v{i}, v{(i + 4) % 64}, v{(i + 8) % 64}, v{i}")
# two VGPR sources, same bank / different bank (add)
tests["v_add_f32 d,d,b   banks differ     "] = block(lambda i: f"v_add_f32 v{i}, v{i}, v{(i + 1) % 64}")
tests["v_add_f32 d,d,b   same bank        "] = block(lambda i: f"v_add_f32 v{i}, v{i}, v{(i + 4) % 64}")
tests["v_mov_b32 d,b                      "] = block(lambda i: f"v_mov_b32 v{i}, v{(i + 17) % 64}")
tests["v_xor_b32 d,s,d                    "] = block(lambda i: f"v_xor_b32 v{i}, s20, v{i}")
# packed
tests["v_pk_mul_f32 D,D,S       (1 vsrc64)"] = block(lambda i: f"v_pk_mul_f32 v[{2*(i%32)}:{2*(i%32)+1}], v[{2*(i%32)}:{2*(i%32)+1}], s[20:21]")
tests["v_pk_fma_f32 D,B,S,D     (2 vsrc64)"] = block(lambda i: f"v_pk_fma_f32 v[{2*(i%32)}:{2*(i%32)+1}], v[{2*((i+9)%32)}:{2*((i+9)%32)+1}], s[20:21], v[{2*(i%32)}:{2*(i%32)+1}]")
tests["v_pk_fma_f32 D,B,S,D opsel swap neg"] = block(lambda i: f"v_pk_fma_f32 v[{2*(i%32)}:{2*(i%32)+1}], v[{2*((i+9)%32)}:{2*((i+9)%32)+1}], s[20:21], v[{2*(i%32)}:{2*(i%32)+1}] op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]")
tests["v_pk_fma_f32 D,A,B,D     (3 vsrc64)"] = block(lambda i: f"v_pk_fma_f32 v[{2*(i%32)}:{2*(i%32)+1}], v[{2*((i+9)%32)}:{2*((i+9)%32)+1}], v[{2*((i+5)%32)}:{2*((i+5)%32)+1}], v[{2*(i%32)}:{2*(i%32)+1}]")
tests["v_pk_add_f32 D,D,B       (2 vsrc64)"] = block(lambda i: f"v_pk_add_f32 v[{2*(i%32)}:{2*(i%32)+1}], v[{2*(i%32)}:{2*(i%32)+1}], v[{2*((i+9)%32)}:{2*((i+9)%32)+1}]")
tests["v_mov_b64 D,B                      "] = block(lambda i: f"v_mov_b64 v[{2*(i%32)}:{2*(i%32)+1}], v[{2*((i+9)%32)}:{2*((i+9)%32)+1}]")
# the candidate gate bodies: real 2x2 on (p,q) with a temp: t=b*q; q=d*q; q=c*p+q; p=a*p+t   (4 ops per rotation)
def rot4(i):
    k = i // 4; p, q, t = 2 * (k % 16), 2 * (k % 16) + 1, 64 + (k % 16)
    return [f"v_mul_f32 v{t}, s21, v{q}", f"v_mul_f32 v{q}, s20, v{q}", f"v_fma_f32 v{q}, -s21, v{p}, v{q}", f"v_fma_f32 v{p}, s20, v{p}, v{t}"][i % 4]
tests["rot4 (mul,mul,fma,fma) dependent    "] = block(rot4)
def rot4i(i):  # two rotations interleaved
    g = i // 8; w = i % 8; k = 2 * g + (w % 2); st = w // 2
    p, q, t = 2 * (k % 16), 2 * (k % 16) + 1, 64 + (k % 16)
    return [f"v_mul_f32 v{t}, s21, v{q}", f"v_mul_f32 v{q}, s20, v{q}", f"v_fma_f32 v{q}, -s21, v{p}, v{q}", f"v_fma_f32 v{p}, s20, v{p}, v{t}"][st]
tests["rot4 interleaved x2                 "] = block(rot4i)
def shear3(i):  # x += u y; y += v x; x += u y
    g = i // 6; w = i % 6; k = 2 * g + (w % 2); st = w // 2
    p, q = 2 * (k % 16), 2 * (k % 16) + 1
    return [f"v_fmac_f32 v{p}, s20, v{q}", f"v_fmac_f32 v{q}, s21, v{p}", f"v_fmac_f32 v{p}, s20, v{q}"][st]
tests["shear3 (fmac x3) interleaved x2     "] = [shear3(i) for i in range(60)]
def pkrot(i):  # complex pair (X,Y) rx-like: T=-is*Y; Y=c*Y; Y+=-is*X; X=c*X+T   4 pk ops per pair
    g = i // 8; w = i % 8; k = 2 * g + (w % 2); st = w // 2
    X = f"v[{4*(k%8)}:{4*(k%8)+1}]"; Y = f"v[{4*(k%8)+2}:{4*(k%8)+3}]"; T = f"v[{64+2*(k%8)}:{64+2*(k%8)+1}]"
    return [f"v_pk_mul_f32 {T}, {Y}, s[20:21] op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]",
            f"v_pk_mul_f32 {Y}, {Y}, s[20:21] op_sel_hi:[1,0]",
            f"v_pk_fma_f32 {Y}, {X}, s[20:21], {Y} op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]",
            f"v_pk_fma_f32 {X}, {X}, s[20:21], {T} op_sel_hi:[1,0,1]"][st]
tests["pk rx pair (4 pk ops) interleaved x2"] = block(pkrot)

print("// GENERATED by gen_operand_forms.py -- do not edit.")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>")
print("#define ITERS 2048")
clob = ", ".join(f'"v{i}"' for i in range(96)) + ', "s20", "s21"'
names = list(tests)
for ti, name in enumerate(names):
    body = "\\n\\t".join(tests[name])
    print(f"template <int LDSB> __global__ __launch_bounds__(256) void k{ti}(float* out, float a, float b) {{")
    print("  extern __shared__ float sm[];")
    print("  if (LDSB && a == 12345.f) sm[threadIdx.x] = b;")
    print('  asm volatile("s_mov_b32 s20, %0\\n\\ts_mov_b32 s21, %1" :: "s"(a), "s"(b) : "s20", "s21");')
    for i in range(96):
        pass
    init = "\\n\\t".join(f"v_mov_b32 v{i}, 1.0" for i in range(96))
    print(f'  asm volatile("{init}" ::: {clob});')
    print("  for (int it = 0; it < ITERS; ++it) {")
    print(f'    asm volatile("{body}" ::: {clob});')
    print("  }")
    print('  float r; asm volatile("v_add_f32 %0, v0, v1" : "=v"(r) :: ' + clob + ");")
    print("  if (r == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = r;")
    print("}")
print("""
template <typename K> static void run(const char* name, K kern, int ninst, size_t lds, int wps) {
  float* d; hipMalloc(&d, 4 << 20);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = 256 * 8;  // 8 workgroups (of 4 waves) per CU in total, run wps/... at a time
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, 0.999f, 0.01f);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, 0.999f, 0.01f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  // every SIMD executes blocks*4/1024 waves in sequence-equivalents: total wave-instructions per SIMD
  const double winst = (double)blocks * 4 / 1024.0 * ITERS * ninst;
  printf("%-40s wps=%d  %8.3f ms  %6.2f ns/winst/SIMD  = %5.2f cyc @2.4GHz  %5.2f cyc @2.1GHz\\n", name, wps, ms,
         ms * 1e6 / winst, ms * 1e6 / winst * 2.4, ms * 1e6 / winst * 2.1);
  hipFree(d);
}
int main(int argc, char** argv) {""")
for ti, name in enumerate(names):
    ni = len(tests[name])
    # 8 waves/SIMD (no LDS), 2 waves/SIMD (64 KiB LDS per WG), 1 wave/SIMD (128 KiB)
    print(f'  run("{name}", k{ti}<0>, {ni}, 0, 8);')
    print(f'  run("{name}", k{ti}<1>, {ni}, 64 * 1024, 2);')
    print(f'  run("{name}", k{ti}<1>, {ni}, 128 * 1024, 1);')
print("  return 0;\n}")
