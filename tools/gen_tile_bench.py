#!/usr/bin/env python3
"""Writes tools/tile_bench.hip: timing of one tile step of the pair search (5 i-particles in SGPRs against the 64 staged
particles a wave holds in VGPRs, the next tile's ds_read_b128 in flight) for several ways of recording the accepted
pairs.  Instruction streams are laid out as hipcc schedules search_group: the five tests interleaved phase by phase.
Timing only -- registers hold arbitrary values.  usage: gen_tile_bench.py > tools/tile_bench.hip"""
G = 5
XI = lambda k: f"s{40 + 4 * k}"
YI = lambda k: f"s{41 + 4 * k}"
ZI = lambda k: f"s{42 + 4 * k}"
GI = lambda k: f"s{43 + 4 * k}"
RC2 = "s60"
M = lambda k: f"s[{64 + 2 * k}:{65 + 2 * k}]"    # in-range masks
U = lambda k: f"s[{76 + 2 * k}:{77 + 2 * k}]"    # upper masks
CNT = lambda k: f"s{90 + k}"
BITS = lambda k: f"v{60 + k}"
T = lambda k, c: f"v{70 + 4 * k + c}"


def math(tile):
    x, y, z = (f"v{tile + c}" for c in range(3))
    o = []
    for k in range(G):
        o += [f"v_subrev_f32 {T(k,0)}, {XI(k)}, {x}", f"v_subrev_f32 {T(k,1)}, {YI(k)}, {y}", f"v_subrev_f32 {T(k,2)}, {ZI(k)}, {z}"]
    for k in range(G):
        o += [f"v_mul_f32 {T(k,0)}, {T(k,0)}, {T(k,0)}", f"v_mul_f32 {T(k,1)}, {T(k,1)}, {T(k,1)}", f"v_mul_f32 {T(k,2)}, {T(k,2)}, {T(k,2)}"]
    for k in range(G):
        o += [f"v_add_f32 {T(k,0)}, {T(k,0)}, {T(k,1)}"]
    for k in range(G):
        o += [f"v_add_f32 {T(k,0)}, {T(k,0)}, {T(k,2)}"]
    return o


def tail(kind, tile):
    gid = f"v{tile + 3}"
    o = []
    if kind == "math":
        pass
    elif kind in ("classic", "classic_nocount"):
        for k in range(G):
            o += [f"v_cmp_nlt_f32 {M(k)}, {RC2}, {T(k,0)}", f"v_cmp_lt_i32 {U(k)}, {GI(k)}, {gid}"]
        for k in range(G):
            o += [f"s_and_b64 {M(k)}, {M(k)}, {U(k)}"]
        for k in range(G):
            o += [f"v_addc_co_u32 {BITS(k)}, s[88:89], {BITS(k)}, {BITS(k)}, {M(k)}"]
            if kind == "classic":
                o += [f"s_bcnt1_i32_b64 s87, {M(k)}", f"s_add_i32 {CNT(k)}, {CNT(k)}, s87"]
    elif kind == "vbits":
        for k in range(G):
            o += [f"v_sub_f32 {T(k,0)}, {RC2}, {T(k,0)}", f"v_subrev_u32 {T(k,1)}, {GI(k)}, {gid}"]
        for k in range(G):
            o += [f"v_or_b32 {T(k,0)}, {T(k,0)}, {T(k,1)}"]
        for k in range(G):
            o += [f"v_alignbit_b32 {BITS(k)}, {BITS(k)}, {T(k,0)}, 31"]
    elif kind == "onecmp":  # sign trick, then ONE compare into vcc and the add-with-carry straight from vcc (e32 forms)
        for k in range(G):
            o += [f"v_sub_f32 {T(k,0)}, {RC2}, {T(k,0)}", f"v_subrev_u32 {T(k,1)}, {GI(k)}, {gid}"]
        for k in range(G):
            o += [f"v_or_b32 {T(k,0)}, {T(k,0)}, {T(k,1)}"]
        for k in range(G):
            o += [f"v_cmp_le_i32 vcc, 0, {T(k,0)}", f"v_addc_co_u32 {BITS(k)}, vcc, {BITS(k)}, {BITS(k)}, vcc"]
    elif kind == "execmask":  # id compare into EXEC, range compare under it, restore, add-with-carry
        for k in range(G):
            o += [f"v_cmpx_lt_i32 {GI(k)}, {gid}", f"v_cmp_nlt_f32 vcc, {RC2}, {T(k,0)}", "s_mov_b64 exec, -1",
                  f"v_addc_co_u32 {BITS(k)}, vcc, {BITS(k)}, {BITS(k)}, vcc"]
    elif kind == "full_classic":  # full list, NOSELF: no id compare
        for k in range(G):
            o += [f"v_cmp_nlt_f32 {M(k)}, {RC2}, {T(k,0)}"]
        for k in range(G):
            o += [f"v_addc_co_u32 {BITS(k)}, s[88:89], {BITS(k)}, {BITS(k)}, {M(k)}", f"s_bcnt1_i32_b64 s87, {M(k)}",
                  f"s_add_i32 {CNT(k)}, {CNT(k)}, s87"]
    elif kind == "full_vcc":  # same through vcc, e32 encodings, count left to the end
        for k in range(G):
            o += [f"v_cmp_nlt_f32 vcc, {RC2}, {T(k,0)}", f"v_addc_co_u32 {BITS(k)}, vcc, {BITS(k)}, {BITS(k)}, vcc"]
    elif kind == "full_vbits":
        for k in range(G):
            o += [f"v_sub_f32 {T(k,0)}, {RC2}, {T(k,0)}"]
        for k in range(G):
            o += [f"v_alignbit_b32 {BITS(k)}, {BITS(k)}, {T(k,0)}, 31"]
    elif kind == "min_shift":  # fold both conditions into one float: r2' = max(r2, id term)?  timing of a v_max_f32 + 1 cmp + addc
        for k in range(G):
            o += [f"v_subrev_u32 {T(k,1)}, {GI(k)}, {gid}", f"v_sub_f32 {T(k,0)}, {RC2}, {T(k,0)}"]
        for k in range(G):
            o += [f"v_or_b32 {T(k,0)}, {T(k,0)}, {T(k,1)}"]
        for k in range(G):
            o += [f"v_cmp_le_i32 {M(k)}, 0, {T(k,0)}"]
        for k in range(G):
            o += [f"v_addc_co_u32 {BITS(k)}, s[88:89], {BITS(k)}, {BITS(k)}, {M(k)}"]
    else:
        raise SystemExit(kind)
    return o


KINDS = ["math", "classic", "classic_nocount", "vbits", "onecmp", "execmask", "min_shift", "full_classic", "full_vcc", "full_vbits"]


def step(kind, tile, other):
    """one tile step: issue the ds_read of the tile after next into `other`... (ping-pong as in search_group)"""
    o = [f"ds_read_b128 v[{other}:{other + 3}], v56", "s_waitcnt lgkmcnt(1)"]
    o += math(tile) + tail(kind, tile)
    o += ["v_add_u32 v56, 0x400, v56", "v_and_b32 v56, 0x3fff, v56"]
    return o


def body(kind):
    lines = step(kind, 46, 50) + step(kind, 50, 46)
    return "\n".join('               "' + l + '\\n"' for l in lines)


clob = ", ".join(f'"v{i}"' for i in list(range(46, 58)) + list(range(60, 66)) + list(range(70, 90))) + ", " + \
       ", ".join(f'"s{i}"' for i in range(64, 96)) + ', "vcc", "scc"'
print("// GENERATED by tools/gen_tile_bench.py -- do not edit.  Tile-step timing probe for gfx950 (timing only).")
print("// Build: hipcc --offload-arch=gfx950 -O3 -o tools/tile_bench tools/tile_bench.hip")
print('#include <hip/hip_runtime.h>\n#include <algorithm>\n#include <cstdio>\n#include <cstdlib>\n#include <vector>')
print('#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)')
print("template <int KIND> __global__ void __launch_bounds__(256) k(float* out, unsigned long long* stamps, int iters) {")
print("  extern __shared__ float lds[];")
print("  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0f + i;")
print("  __syncthreads();")
print("  unsigned long long t0, t1;")
print('  asm volatile("v_lshlrev_b32 v56, 4, %0\\n s_mov_b64 exec, -1\\n" :: "v"(threadIdx.x & 63) : "v56");')
for r in range(40, 64):
    pass
print('  asm volatile("s_mov_b32 s40, 1.0\\n s_mov_b32 s41, 2.0\\n s_mov_b32 s42, 0.5\\n s_mov_b32 s43, 7\\n s_mov_b32 s44, 1.0\\n s_mov_b32 s45, 2.0\\n s_mov_b32 s46, 0.5\\n s_mov_b32 s47, 9\\n"')
print('               "s_mov_b32 s48, 1.0\\n s_mov_b32 s49, 2.0\\n s_mov_b32 s50, 0.5\\n s_mov_b32 s51, 11\\n s_mov_b32 s52, 1.0\\n s_mov_b32 s53, 2.0\\n s_mov_b32 s54, 0.5\\n s_mov_b32 s55, 13\\n"')
print('               "s_mov_b32 s56, 1.0\\n s_mov_b32 s57, 2.0\\n s_mov_b32 s58, 0.5\\n s_mov_b32 s59, 15\\n s_mov_b32 s60, 4.0\\n"')
print('               ::: "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60");')
print('  asm volatile("s_memtime %0\\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");')
print("  for (int i = 0; i < iters; i++) {")
for n, kind in enumerate(KINDS):
    print(f"    if (KIND == {n}) {{  // {kind}")
    print("      asm volatile(")
    print(body(kind))
    print(f"               ::: {clob}, \"memory\");")
    print("    }")
print("  }")
print('  asm volatile("s_waitcnt lgkmcnt(0)\\n s_memtime %0\\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");')
print("  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;")
print("  if (iters < 0) out[threadIdx.x] = lds[threadIdx.x];")
print("}")
print("""template <int KIND> void run(const char* name, unsigned long long* stamps_d, float* out) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, iters = 2000;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-18s", name);
  for (int bpc : {8, 7, 6, 4, 2, 1}) {  // workgroups of 4 waves per CU = waves per SIMD
    const size_t lds = (size_t)(160 * 1024 / bpc) & ~(size_t)1023;
    const int blocks = cus * bpc, nw = blocks * 4;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, out, stamps_d, 50);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), lds, 0, out, stamps_d, iters);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(nw);
    CHK(hipMemcpy(st.data(), stamps_d, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    std::sort(st.begin(), st.end());
    // cycles per wave-test per SIMD: a wave's elapsed cycles / (tests it ran) / waves sharing the SIMD
    printf(" | w%d %6.2f", bpc, (double)st[nw / 2] / ((double)iters * 10) / bpc);
  }
  printf("\\n");
}
int main() {
  float* out; unsigned long long* stamps;
  CHK(hipMalloc(&out, 4096)); CHK(hipMalloc(&stamps, sizeof(unsigned long long) * 256 * 8 * 4));
  printf("cycles (s_memtime) per wave-test per SIMD; one tile step = 5 tests + ds_read_b128 of the next tile; wN = N waves per SIMD\\n");""")
for n, kind in enumerate(KINDS):
    print(f'  run<{n}>("{kind}", stamps, out);')
print("  return 0;\n}")
