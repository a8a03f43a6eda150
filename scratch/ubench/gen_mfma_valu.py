"""Generates scratch/ubench/mfma_valu_overlap.hip: how many VALU instructions hide under one v_mfma_f32_32x32x16_f16 on gfx950,
by accumulator register file (arch VGPR vs AccVGPR), VALU kind (v_fma_f32 / v_exp_f32 / v_cvt_pk_f16_f32) and waves per SIMD."""
import itertools
NVS = [0, 2, 4, 6, 8, 12]
KINDS = {"fma": "v_fma_f32 {r}, {r}, {r}, {r}", "exp": "v_exp_f32 {r}, {r}", "cvt": "v_cvt_pk_f16_f32 {r}, {r}, {r}",
         "max3": "v_max3_f32 {r}, {r}, {r}, {r}"}
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>',
       'typedef float v16f __attribute__((ext_vector_type(16)));', 'typedef _Float16 v8h __attribute__((ext_vector_type(8)));', '']
names = []
for form, kind, nv in itertools.product("va", KINDS, NVS):
    if nv == 0 and kind != "fma":
        continue
    name = f"k_{form}_{kind}_{nv}"
    names.append((name, form, kind, nv))
    body = []
    for m in range(2):                       # two MFMAs on independent accumulators per loop trip, nv VALU behind each
        body.append(f"v_mfma_f32_32x32x16_f16 %{m}, %2, %3, %{m}")
        for i in range(nv):
            body.append(KINDS[kind].format(r=f"%{4 + (m * nv + i) % 8}"))
    asm = "\\n\\t".join(body)
    src.append(f'''__global__ __launch_bounds__(256) void {name}(float* out, int iters) {{
  v16f a0, a1; v8h fa, fb; float x[8];
  for (int i = 0; i < 16; ++i) {{ a0[i] = threadIdx.x * 1e-3f; a1[i] = i; }}
  for (int i = 0; i < 8; ++i) {{ fa[i] = (_Float16)(threadIdx.x * 1e-2f); fb[i] = (_Float16)0.5f; x[i] = 0.25f + i * 0.01f; }}
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("{asm}" : "+{form}"(a0), "+{form}"(a1) : "v"(fa), "v"(fb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 22))[0] = t1 - t0;
}}
''')
src.append('''int main(int argc, char** argv) {
  const int iters = 4000; float* out; hipMalloc(&out, (1 << 22) * 4 + 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-16s %5s %9s %12s %12s\\n", "kernel", "w/SIMD", "us", "ns/MFMA/SIMD", "clk/MFMA");
  for (int occ = 1; occ <= 3; ++occ) {''')
for name, form, kind, nv in names:
    src.append(f'''    {{ {name}<<<256 * occ, 256>>>(out, 10); hipDeviceSynchronize(); hipEventRecord(e0); {name}<<<256 * occ, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); long long clk; hipMemcpy(&clk, out + (1 << 22), 8, hipMemcpyDeviceToHost);
      printf("%-16s %5d %9.1f %12.2f %12.1f\\n", "{name}", occ, ms * 1e3, ms * 1e6 / (iters * 2.0 * occ), (double)clk / (iters * 2.0 * occ)); }}''')
src.append('  }\n  return 0;\n}')
open(__file__.replace("gen_mfma_valu.py", "mfma_valu_overlap.hip"), "w").write("\n".join(src))
