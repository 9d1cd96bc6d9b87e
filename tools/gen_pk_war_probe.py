"""Generates the register-exact bodies of k_pk_war_probe (csrc/dd_tools.hip, tools/pkfma_war_repro.hip): the packed-FP32 P.V step with the registers
fixed by hand.  Variants 0-2: round 5's first probe (the LDS-fed pair as src0); variants 3-5: the COMPILER's instruction group of k_pv_probe
replicated (the LDS-fed pair as src1 with op_sel broadcast, a v_mov_b32 into the pair's low register between the packed ops):
  3  the group as compiled: ..., v_mov_b32 v40, v43, ..., then ds_read_b128 v[40:43] right behind the packed ops
  4  the same with sixteen wait states before the re-load
  5  the same without the v_mov (the fourth probability read as the high half of v[42:43])
    python tools/gen_pk_war_probe.py            # prints the C++ of the kernel template"""
NU = 32


def pk8_src0(q, lo):
    base = 0 if lo else 8
    sel = ['op_sel_hi:[0,1,1]', 'op_sel:[1,0,0] op_sel_hi:[1,1,1]']
    out = ''
    for r in range(4):
        src = f'v[{q + 2 * (r // 2)}:{q + 2 * (r // 2) + 1}]'
        for w in range(2):
            acc = f'%{base + 2 * r + w}'
            out += f'          "v_pk_fma_f32 {acc}, {src}, %{16 + w}, {acc} {sel[r % 2]}\\n"\n'
    return out


def pk8_compiler(q, lo, with_mov):
    """hipcc's group: p as src1; rows 0, 1 from v[q:q+1] (low / high), row 2 from v[q+2:q+3] low, row 3 via v_mov into v[q] (or high of q+2)."""
    base = 0 if lo else 8
    lo_sel, hi_sel = 'op_sel_hi:[1,0,1]', 'op_sel:[0,1,0]'
    out = ''
    def two(r, src, sel):
        return ''.join(f'          "v_pk_fma_f32 %{base + 2 * r + w}, %{16 + w}, {src}, %{base + 2 * r + w} {sel}\\n"\n' for w in range(2))
    out += two(0, f'v[{q}:{q + 1}]', lo_sel)
    out += two(1, f'v[{q}:{q + 1}]', hi_sel)
    if with_mov:
        out += f'          "v_mov_b32 v{q}, v{q + 3}\\n"\n'
    out += two(2, f'v[{q + 2}:{q + 3}]', lo_sel)
    out += two(3, f'v[{q}:{q + 1}]', lo_sel) if with_mov else two(3, f'v[{q + 2}:{q + 3}]', hi_sel)
    return out


def single_quad(body, nop):
    out = '          "ds_read_b128 v[40:43], %18\\n"\n'
    for u in range(NU):
        out += '          "s_waitcnt lgkmcnt(0)\\n"\n' + body(40, u % 2 == 0)
        if nop:
            out += '          "s_nop 7\\n"\n          "s_nop 7\\n"\n'
        if u + 1 < NU:
            out += f'          "ds_read_b128 v[40:43], %18 offset:{16 * (u + 1)}\\n"\n'
    return out


def rotating():
    quads = [40, 44, 48, 52, 56, 60]
    out = f'          "ds_read_b128 v[{quads[0]}:{quads[0] + 3}], %18\\n"\n          "ds_read_b128 v[{quads[1]}:{quads[1] + 3}], %18 offset:16\\n"\n'
    for u in range(NU):
        out += f'          "s_waitcnt lgkmcnt({1 if u + 1 < NU else 0})\\n"\n'
        if u + 2 < NU:
            q = quads[(u + 2) % 6]
            out += f'          "ds_read_b128 v[{q}:{q + 3}], %18 offset:{16 * (u + 2)}\\n"\n'
        out += pk8_src0(quads[u % 6], u % 2 == 0)
    return out + '          "s_waitcnt lgkmcnt(0)\\n"\n'


CLOB = ', '.join(f'"v{i}"' for i in range(40, 64))
OUTS = ', '.join(f'"+v"(a[{i}])' for i in range(16))


def blk(txt):
    return f'''      asm volatile(
{txt}          : {OUTS}
          : "v"(va), "v"(vb), "v"(addr)
          : {CLOB}, "memory");
'''


VARIANTS = [single_quad(pk8_src0, False), single_quad(pk8_src0, True), rotating(),
            single_quad(lambda q, lo: pk8_compiler(q, lo, True), False), single_quad(lambda q, lo: pk8_compiler(q, lo, True), True),
            single_quad(lambda q, lo: pk8_compiler(q, lo, False), False)]


def kernel():
    chain = ''
    for i, v in enumerate(VARIANTS):
        chain += ('    if constexpr (VARIANT == 0) {\n' if i == 0 else (f'    }} else if constexpr (VARIANT == {i}) {{\n' if i + 1 < len(VARIANTS) else '    } else {\n')) + blk(v)
    chain += '    }\n'
    return f'''template <int VARIANT>
__global__ __launch_bounds__(256) void k_pk_war_probe(int iters, uint32_t salt, unsigned int* errors) {{
  extern __shared__ __align__(16) float pf[];
  float* p_sh = pf + 3072;                             // [64 keys][8 rows], as in k_pv_probe
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t me = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u) ^ salt;
  dd_f32x2 va = {{0.5f + (float)(me & 255u) * (1.0f / 256.0f), -0.25f + (float)((me >> 8) & 127u) * (1.0f / 128.0f)}};
  dd_f32x2 vb = {{1.0f - (float)((me >> 15) & 63u) * (1.0f / 64.0f), 0.125f + (float)((me >> 21) & 31u) * (1.0f / 32.0f)}};
  unsigned int bad = 0;
  for (int it = 0; it < iters; ++it) {{
    for (int r = wave; r < 8; r += 4) p_sh[lane * 8 + r] = (float)((lane * 8 + r + it * 13 + (int)(salt & 31u)) & 511) * (1.0f / 512.0f);
    __syncthreads();
    // packed: a[2 r + w] += p[key][r] * (w ? vb : va) for r = 0..7 over the wave's 16 keys; a key's eight probabilities = two 16-byte units
    // (rows 0-3 / 4-7), eight packed ops per unit as in the tile pass
    dd_f32x2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (dd_f32x2){{0.f, 0.f}};
    const uint32_t addr = (uint32_t)(uintptr_t)(p_sh + wave * 16 * 8);      // LDS byte address of the wave's first key
{chain}    // scalar reference: the same sums, same order per accumulator
    float ref[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) ref[i][0] = ref[i][1] = 0.f;
#pragma unroll 2
    for (int key = 0; key < 16; ++key) {{
      const float* pr = &p_sh[(wave * 16 + key) * 8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {{
        float pv = pr[r];
        asm volatile("" : "+v"(pv));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * r][0]) : "v"(pv), "v"(va.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * r][1]) : "v"(pv), "v"(va.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * r + 1][0]) : "v"(pv), "v"(vb.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref[2 * r + 1][1]) : "v"(pv), "v"(vb.y));
      }}
    }}
#pragma unroll
    for (int i = 0; i < 16; ++i)
      bad += (__float_as_uint(a[i].x) != __float_as_uint(ref[i][0])) | (__float_as_uint(a[i].y) != __float_as_uint(ref[i][1]));
    va = va * 0.9995f + 0.0003f;
    vb = vb * 1.0002f - 0.0001f;
    __syncthreads();
  }}
  if (bad) atomicAdd(errors + VARIANT, bad);
}}
'''


if __name__ == "__main__":
    print(kernel())
