// ViT vision towers on own kernels (the vision front-end): CLIP ViT-L/14 + LLaVA projector, and
// InstructBLIP's EVA ViT-g/14 (1408 wide, 16 heads of 88 — padded to a pitch of 96 for the matrix-core attention —, no
// pre-LayerNorm, patch bias, post-LayerNorm over all 257 tokens: what models/instructblip.py:607-612 gets from
// `self.vision_model(...)`).
// Replaces, for LLaVA-1.5 / LLaVA-NeXT tiles, what the reference runs through third-party modules at
// models/llava.py:233-246 (vision_tower(pixel_values, output_hidden_states=True).hidden_states[-2][:, 1:] ->
// multi_modal_projector) and models/llavanext.py:409-417.
// Same numerics policy as the LM: bf16 weights, fp32 activations carried as bf16 hi/lo planes through the MFMA GEMM
// (k_gemm, dd_lm_kernels.hip), fp32 LayerNorm / softmax / residual stream.
#include <math.h>

#include <vector>

#include "dd_lm_kernels.h"

// ---- tensor ids (HF CLIPVisionModel / LlavaMultiModalProjector parameter names) -----------------------------------
// see include/dropdec.h

__global__ void k_pad_head_cols(const uint16_t* __restrict__ src, int d, int hd, int hp, int H, uint16_t* __restrict__ dst);

struct VitLayer {
  u32x4_t *wqkv, *wo, *wfc1, *wfc2;
  float *bqkv, *bo, *bfc1, *bfc2, *ln1w, *ln1b, *ln2w, *ln2b;
};

struct dd_vit {
  dd_vit_config cfg;
  int T, Tc, P, d, dff, H, hd, hp, Kp, Sp, proj;   // tokens (patches+1), padded tokens, patches, hidden, mlp, heads, head dim, head pitch, padded patch K
  int no_pre_ln = 0, post_ln = 0, keep_cls = 0;
  std::vector<void*> allocs;
  size_t bytes = 0;
  std::vector<VitLayer> lw;
  u32x4_t *wpatch = nullptr, *wp1 = nullptr, *wp2 = nullptr;
  float *cls = nullptr, *pos = nullptr, *prew = nullptr, *preb = nullptr, *bp1 = nullptr, *bp2 = nullptr;
  float *bpatch = nullptr, *postw = nullptr, *postb = nullptr;
  uint16_t* wo_pad = nullptr;   // staging for o_proj weights with head columns spread to the head pitch
  // scratch
  float *x = nullptr, *q = nullptr, *kt = nullptr, *v = nullptr;
  uint16_t *a_hi = nullptr, *a_lo = nullptr, *b_hi = nullptr, *b_lo = nullptr;
  // scratch of the batched forward (several images as one matrix), allocated on the first batch: bcap images of Tp rows
  int bcap = 0, Tp = 0;
  float *bx = nullptr, *bq = nullptr, *bkt = nullptr, *bv = nullptr;
  uint16_t *ba_hi = nullptr, *ba_lo = nullptr, *bb_hi = nullptr, *bb_lo = nullptr;
  SeqTab* btab = nullptr;
};

template <typename T>
static int valloc(dd_vit* h, T** p, size_t n) {
  void* q = nullptr;
  size_t b = n * sizeof(T);
  if (b == 0) b = 16;
  if (hipMalloc(&q, b) != hipSuccess) {
    dd_set_error("dd_vit: hipMalloc(%zu) failed", b);
    return DD_ENOMEM;
  }
  (void)hipMemset(q, 0, b);
  h->allocs.push_back(q);
  h->bytes += b;
  *p = (T*)q;
  return DD_OK;
}
#define VA(ptr, n)                                   \
  do {                                               \
    int rc__ = valloc(h, &(ptr), (size_t)(n));       \
    if (rc__ != DD_OK) {                             \
      dd_vit_destroy(h);                             \
      return rc__;                                   \
    }                                                \
  } while (0)
#define RC(expr)                    \
  do {                              \
    int rc__ = (expr);              \
    if (rc__ != DD_OK) return rc__; \
  } while (0)

extern "C" int dd_vit_destroy(dd_vit* h) {
  if (!h) return DD_OK;
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return DD_OK;
}

extern "C" int dd_vit_create(const dd_vit_config* c, dd_vit** out) {
  DD_REQUIRE(c && out, "dd_vit_create: null argument");
  DD_REQUIRE(c->image_size % c->patch_size == 0, "dd_vit_create: image %d not a multiple of patch %d", c->image_size, c->patch_size);
  DD_REQUIRE(c->hidden_size % 64 == 0 && c->intermediate_size % 64 == 0, "dd_vit_create: hidden/intermediate must be multiples of 64");
  DD_REQUIRE(c->hidden_size % c->num_heads == 0, "dd_vit_create: heads");
  int hd = c->hidden_size / c->num_heads;
  DD_REQUIRE(hd == 64 || hd == 88, "dd_vit_create: head_dim must be 64 (CLIP ViT-L/14) or 88 (EVA ViT-g/14), got %d", hd);
  DD_REQUIRE(c->proj_dim == 0 || c->proj_dim % 64 == 0, "dd_vit_create: proj_dim must be a multiple of 64");
  DD_REQUIRE(c->num_layers >= 1, "dd_vit_create: num_layers");
  dd_vit* h = new dd_vit();
  h->cfg = *c;
  int g = c->image_size / c->patch_size;
  h->P = g * g, h->T = h->P + 1, h->Tc = (h->T + 63) / 64 * 64;
  h->d = c->hidden_size, h->dff = c->intermediate_size, h->H = c->num_heads, h->hd = hd, h->proj = c->proj_dim;
  h->hp = (hd + 31) / 32 * 32;                      // 64 -> 64, 88 -> 96
  h->no_pre_ln = c->flags & DD_VIT_NO_PRE_LN ? 1 : 0, h->post_ln = c->flags & DD_VIT_POST_LN ? 1 : 0, h->keep_cls = c->flags & DD_VIT_KEEP_CLASS ? 1 : 0;
  DD_REQUIRE(!(h->proj && h->keep_cls), "dd_vit_create: the projector path drops the class token");
  int kp = 3 * c->patch_size * c->patch_size;
  h->Kp = (kp + 63) / 64 * 64, h->Sp = h->Kp / 32;
  const int d = h->d, dff = h->dff;
  h->lw.resize(c->num_layers);
  for (auto& w : h->lw) {
    VA(w.wqkv, (size_t)(3 * d / 16) * (d / 32) * 64);
    VA(w.wo, (size_t)(d / 16) * (h->H * h->hp / 32) * 64);
    VA(w.wfc1, (size_t)(dff / 16) * (d / 32) * 64);
    VA(w.wfc2, (size_t)(d / 16) * (dff / 32) * 64);
    VA(w.bqkv, 3 * d); VA(w.bo, d); VA(w.bfc1, dff); VA(w.bfc2, d);
    VA(w.ln1w, d); VA(w.ln1b, d); VA(w.ln2w, d); VA(w.ln2b, d);
  }
  VA(h->wpatch, (size_t)(d / 16) * h->Sp * 64);
  VA(h->cls, d); VA(h->pos, (size_t)h->T * d); VA(h->prew, d); VA(h->preb, d);
  VA(h->bpatch, d); VA(h->postw, d); VA(h->postb, d);
  if (h->hp != hd) VA(h->wo_pad, (size_t)d * h->H * h->hp);
  if (h->proj) {
    VA(h->wp1, (size_t)(h->proj / 16) * (d / 32) * 64);
    VA(h->wp2, (size_t)(h->proj / 16) * (h->proj / 32) * 64);
    VA(h->bp1, h->proj); VA(h->bp2, h->proj);
  }
  size_t wmax = (size_t)(dff > h->Kp ? dff : h->Kp);
  if ((size_t)h->proj > wmax) wmax = h->proj;
  const size_t dq = (size_t)h->H * h->hp;           // width of q / K^T / V and of the attention output planes
  size_t amax = d > h->Kp ? d : h->Kp;
  if (dq > amax) amax = dq;
  VA(h->x, (size_t)h->Tc * d);
  VA(h->q, (size_t)h->Tc * dq);
  VA(h->kt, (size_t)h->Tc * dq);
  VA(h->v, (size_t)h->Tc * dq);
  VA(h->a_hi, (size_t)h->Tc * amax);
  VA(h->a_lo, (size_t)h->Tc * amax);
  VA(h->b_hi, (size_t)h->Tc * wmax);
  VA(h->b_lo, (size_t)h->Tc * wmax);
  *out = h;
  return DD_OK;
}

extern "C" int dd_vit_load_tensor(dd_vit* h, int id, int layer, const uint16_t* src, int rows, int cols, int on_device) {
  DD_REQUIRE(h && src, "dd_vit_load_tensor: null argument");
  DD_REQUIRE(id >= DD_VT_PATCH && id <= DD_VT_POST_LN_B, "dd_vit_load_tensor: unknown tensor id %d", id);
  bool per_layer = id >= DD_VT_LN1_W && id <= DD_VT_FC2_B;
  DD_REQUIRE(!per_layer || (layer >= 0 && layer < (int)h->lw.size()), "dd_vit_load_tensor: layer %d out of range", layer);
  const int d = h->d, dff = h->dff;
  size_t n = (size_t)rows * cols;
  const uint16_t* dev = src;
  uint16_t* staging = nullptr;
  if (!on_device) {
    DD_HIP(hipMalloc((void**)&staging, n * 2));
    if (hipMemcpy(staging, src, n * 2, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(staging);
      dd_set_error("dd_vit_load_tensor: H2D copy failed");
      return DD_EHIP;
    }
    dev = staging;
  }
  VitLayer* w = per_layer ? &h->lw[layer] : nullptr;
  int rc = DD_OK;
  auto vec = [&](float* dst, int len) -> int {
    if ((size_t)len != n) {
      dd_set_error("dd_vit_load_tensor: tensor %d expects %d values, got %zu", id, len, n);
      return DD_EINVAL;
    }
    return ddk_bf16_to_f32(dev, dst, len, nullptr);
  };
  auto mat = [&](u32x4_t* dst, int er, int ec, int tile0, int nt_total) -> int {
    if (rows != er || cols != ec) {
      dd_set_error("dd_vit_load_tensor: tensor %d expects %d x %d, got %d x %d", id, er, ec, rows, cols);
      return DD_EINVAL;
    }
    (void)nt_total;
    return ddk_pack_weight(dev, rows, cols, dst, tile0, 1, PACK_PLAIN, er / 16, nullptr);
  };
  switch (id) {
    case DD_VT_PATCH: rc = mat(h->wpatch, d, h->Kp, 0, d / 16); break;   // conv weight flattened [d][3*p*p], zero-padded to Kp
    case DD_VT_CLASS: rc = vec(h->cls, d); break;
    case DD_VT_POS: rc = vec(h->pos, h->T * d); break;
    case DD_VT_PRE_LN_W: rc = vec(h->prew, d); break;
    case DD_VT_PRE_LN_B: rc = vec(h->preb, d); break;
    case DD_VT_LN1_W: rc = vec(w->ln1w, d); break;
    case DD_VT_LN1_B: rc = vec(w->ln1b, d); break;
    case DD_VT_WQ: rc = mat(w->wqkv, d, d, 0, 3 * d / 16); break;
    case DD_VT_WK: rc = mat(w->wqkv, d, d, d / 16, 3 * d / 16); break;
    case DD_VT_WV: rc = mat(w->wqkv, d, d, 2 * d / 16, 3 * d / 16); break;
    case DD_VT_BQ: rc = vec(w->bqkv, d); break;
    case DD_VT_BK: rc = vec(w->bqkv + d, d); break;
    case DD_VT_BV: rc = vec(w->bqkv + 2 * d, d); break;
    case DD_VT_WO:
      if (h->hp == h->hd) {
        rc = mat(w->wo, d, d, 0, d / 16);
      } else {   // out_proj reads the attention output at the head pitch: its columns move to head * pitch + idx, pads are zero
        DD_REQUIRE(rows == d && cols == d, "dd_vit_load_tensor: out_proj expects %d x %d", d, d);
        k_pad_head_cols<<<d, 256>>>(dev, d, h->hd, h->hp, h->H, h->wo_pad);
        rc = ddk_pack_weight(h->wo_pad, d, h->H * h->hp, w->wo, 0, 1, PACK_PLAIN, d / 16, nullptr);
      }
      break;
    case DD_VT_BO: rc = vec(w->bo, d); break;
    case DD_VT_LN2_W: rc = vec(w->ln2w, d); break;
    case DD_VT_LN2_B: rc = vec(w->ln2b, d); break;
    case DD_VT_FC1_W: rc = mat(w->wfc1, dff, d, 0, dff / 16); break;
    case DD_VT_FC1_B: rc = vec(w->bfc1, dff); break;
    case DD_VT_FC2_W: rc = mat(w->wfc2, d, dff, 0, d / 16); break;
    case DD_VT_FC2_B: rc = vec(w->bfc2, d); break;
    case DD_VT_PROJ1_W: DD_REQUIRE(h->proj, "no projector configured"); rc = mat(h->wp1, h->proj, d, 0, h->proj / 16); break;
    case DD_VT_PROJ1_B: DD_REQUIRE(h->proj, "no projector configured"); rc = vec(h->bp1, h->proj); break;
    case DD_VT_PROJ2_W: DD_REQUIRE(h->proj, "no projector configured"); rc = mat(h->wp2, h->proj, h->proj, 0, h->proj / 16); break;
    case DD_VT_PROJ2_B: DD_REQUIRE(h->proj, "no projector configured"); rc = vec(h->bp2, h->proj); break;
    case DD_VT_PATCH_B: rc = vec(h->bpatch, d); break;
    case DD_VT_POST_LN_W: rc = vec(h->postw, d); break;
    case DD_VT_POST_LN_B: rc = vec(h->postb, d); break;
  }
  hipError_t e = hipDeviceSynchronize();
  if (staging) (void)hipFree(staging);
  if (rc != DD_OK) return rc;
  DD_HIP(e);
  return DD_OK;
}

// ---- kernels --------------------------------------------------------------------------------------------------------
// W_o [d][H * hd] -> [d][H * hp] (bf16 bits), head h's columns at h * hp, pad columns zero
__global__ __launch_bounds__(256) void k_pad_head_cols(const uint16_t* __restrict__ src, int d, int hd, int hp, int H, uint16_t* __restrict__ dst) {
  const int row = blockIdx.x;
  for (int c = threadIdx.x; c < H * hp; c += 256) {
    int head = c / hp, idx = c % hp;
    dst[(size_t)row * H * hp + c] = idx < hd ? src[(size_t)row * d + head * hd + idx] : (uint16_t)0;
  }
}
// pixels [3][H][W] fp32 -> im2col rows (one per patch, k = c*p*p + ky*p + kx, zero-padded to Kp) as packed hi/lo planes
// with a row offset of 1 (row 0 is the class token, filled elsewhere)
__global__ __launch_bounds__(256) void k_vit_patchify(const float* __restrict__ px, int img, int p, int Kp, uint16_t* hi,
                                                      uint16_t* lo, int row0 = 0) {
  int patch = blockIdx.x, g = img / p;
  int py = patch / g, pxx = patch % g;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    float v = 0.f;
    if (k < 3 * p * p) {
      int c = k / (p * p), r = k % (p * p), ky = r / p, kx = r % p;
      v = px[((size_t)c * img + py * p + ky) * img + pxx * p + kx];
    }
    uint32_t h, l;
    dd_split_hl(v, h, l);
    size_t o = apack_off(row0 + patch + 1, k, Kp >> 5);
    hi[o] = (uint16_t)h;
    lo[o] = (uint16_t)l;
  }
}
// x[0] = class_embedding + pos[0]; x[r] = patch_embed[r] + pos[r]
__global__ __launch_bounds__(256) void k_vit_embed(float* x, const float* cls, const float* pos, int d) {
  int row = blockIdx.x;
  for (int i = threadIdx.x; i < d; i += 256) {
    float v = row == 0 ? cls[i] : x[(size_t)row * d + i];
    x[(size_t)row * d + i] = v + pos[(size_t)row * d + i];
  }
}
// the same for the tokens of several images back to back (Tp rows per image, T live ones)
__global__ __launch_bounds__(256) void k_vit_embed_batch(float* x, const float* cls, const float* pos, int d, int T, int Tp) {
  const int row = blockIdx.x, r = row % Tp;
  if (r >= T) return;
  for (int i = threadIdx.x; i < d; i += 256) {
    float v = r == 0 ? cls[i] : x[(size_t)row * d + i];
    x[(size_t)row * d + i] = v + pos[(size_t)r * d + i];
  }
}
// LayerNorm (biased variance, fp32): out_f32 (optional, may alias x) and/or packed hi/lo planes at row (row + row_off)
// img_out_rows > 0: a gather over images — output row j takes source row (j / img_out_rows) * img_src_rows + src_row0 + j % img_out_rows
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, int d, const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps, float* out_f32, uint16_t* hi,
                                                   uint16_t* lo, int src_row0, int img_out_rows = 0, int img_src_rows = 0) {
  __shared__ float sh[8];
  int row = blockIdx.x;
  const float* xr = x + (size_t)(img_out_rows ? (row / img_out_rows) * img_src_rows + src_row0 + row % img_out_rows : row + src_row0) * d;
  float s = 0.f;
  for (int i = threadIdx.x; i < d; i += 256) s += xr[i];
  s = dd_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)d;
  float v = 0.f;
  for (int i = threadIdx.x; i < d; i += 256) {
    float t = xr[i] - mean;
    v += t * t;
  }
  v = dd_wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[4 + (threadIdx.x >> 6)] = v;
  __syncthreads();
  float rstd = 1.0f / sqrtf((sh[4] + sh[5] + sh[6] + sh[7]) / (float)d + eps);
  for (int i = threadIdx.x; i < d; i += 256) {
    float y = w ? (xr[i] - mean) * rstd * w[i] + b[i] : xr[i];   // w == nullptr: plain split (projector input)
    if (out_f32) out_f32[(size_t)row * d + i] = y;
    if (hi) {
      uint32_t h, l;
      dd_split_hl(y, h, l);
      size_t o = apack_off(row, i, d >> 5);
      hi[o] = (uint16_t)h;
      lo[o] = (uint16_t)l;
    }
  }
}
// bidirectional attention, head_dim 64, one wave per (head, query): keys on lanes from the transposed K, values with
// 4 keys x 16 d-quads per wave instruction
__global__ __launch_bounds__(256) void k_attn_vit(const float* __restrict__ q, const float* __restrict__ kt,
                                                  const float* __restrict__ v, int T, int Tc, int d, uint16_t* o_hi,
                                                  uint16_t* o_lo) {
  __shared__ __align__(16) float q_sh[4][64];
  __shared__ float p_sh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = blockIdx.x, t = blockIdx.y * 4 + wave;
  const bool live = t < T;
  const int tq = live ? t : T - 1;
  q_sh[wave][lane] = q[(size_t)tq * d + head * 64 + lane];
  __syncthreads();
  const int kg = lane >> 4, dq = lane & 15;   // P.V: key group (4 keys per instruction), d-quad
  float m_run = -INFINITY, l_run = 0.f;
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < T; t0 += 64) {
    int key = t0 + lane;
    bool valid = key < T;
    const float* kb = kt + ((size_t)head * 16 * Tc + (valid ? key : 0)) * 4;
    float s = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) {
      f32x4_t k4 = *(const f32x4_t*)(kb + (size_t)d4 * Tc * 4);
      f32x4_t q4 = *(const f32x4_t*)&q_sh[wave][d4 * 4];
      s += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
    }
    if (!valid) s = -INFINITY;
    float m_new = fmaxf(m_run, dd_wave_max(s));
    float p = valid ? expf(s - m_new) : 0.f;
    float corr = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    l_run = l_run * corr + dd_wave_sum(p);
    acc *= corr;
    m_run = m_new;
    p_sh[wave][lane] = p;
    __builtin_amdgcn_wave_barrier();
    int nkeys = min(64, T - t0);
    const float* vb = v + ((size_t)head * Tc + t0) * 64 + dq * 4;
    for (int k4i = 0; 4 * k4i < nkeys; ++k4i) {
      int kk = 4 * k4i + kg;
      if (kk < nkeys) acc += p_sh[wave][kk] * *(const f32x4_t*)(vb + (size_t)kk * 64);
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o);
    acc.y += __shfl_xor(acc.y, o);
    acc.z += __shfl_xor(acc.z, o);
    acc.w += __shfl_xor(acc.w, o);
  }
  if (live && kg == 0) {
    float inv = 1.0f / l_run;
    uint32_t hh[4], ll[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dd_split_hl(acc[j] * inv, hh[j], ll[j]);
    size_t o = apack_off(t, head * 64 + dq * 4, d >> 5);
    *(u32x2_t*)(o_hi + o) = (u32x2_t){hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16)};
    *(u32x2_t*)(o_lo + o) = (u32x2_t){ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16)};
  }
}

// ---- forward ----------------------------------------------------------------------------------------------------------
// nb images at once: each image's T tokens padded to Tp (whole 128-row blocks) and all of them run through the layers as one
// matrix (the GEMMs of one 577-token image are 5 row blocks: launch-bound at a seventh of the matrix-core rate), attention as
// one launch over the images.  Rows are independent everywhere else, so the result equals nb single-image forwards bit for bit.
static int vit_forward_batch(dd_vit* h, const float* pixels, int nb, float* out, hipStream_t st) {
  const int d = h->d, dff = h->dff, T = h->T, P = h->P, img = h->cfg.image_size;
  const size_t dq = (size_t)h->H * h->hp;
  if (h->bcap < nb) {
    DD_HIP(hipDeviceSynchronize());
    const int cap = 16;
    h->Tp = (T + 127) / 128 * 128;
    const size_t M = (size_t)cap * h->Tp;
    size_t wmax = (size_t)(dff > h->Kp ? dff : h->Kp);
    if ((size_t)h->proj > wmax) wmax = h->proj;
    size_t amax = d > h->Kp ? d : h->Kp;
    if (dq > amax) amax = dq;
    if (valloc(h, &h->bx, M * d) || valloc(h, &h->bq, M * dq) || valloc(h, &h->bkt, M * dq) || valloc(h, &h->bv, M * dq) ||
        valloc(h, &h->ba_hi, M * amax) || valloc(h, &h->ba_lo, M * amax) || valloc(h, &h->bb_hi, M * wmax) || valloc(h, &h->bb_lo, M * wmax) ||
        valloc(h, &h->btab, 1))
      return DD_ENOMEM;
    h->bcap = cap;
  }
  const int Tp = h->Tp, M = nb * Tp;
  const size_t img_kv = (size_t)Tp * dq;                 // floats of one image's K^T (V) block
  {
    SeqTab tab;
    memset(&tab, 0, sizeof(tab));
    for (int i = 0; i < nb; ++i) tab.T[i] = T, tab.kc[i] = h->bkt + (size_t)i * img_kv, tab.vc[i] = h->bv + (size_t)i * img_kv;
    RC(ddk_put_seq_tab(tab, h->btab, st));
  }
  DD_HIP(hipMemsetAsync(h->ba_hi, 0, (size_t)M * h->Kp * 2, st));   // class-token and padding rows of the patch operand: zeros
  DD_HIP(hipMemsetAsync(h->ba_lo, 0, (size_t)M * h->Kp * 2, st));
  for (int i = 0; i < nb; ++i) {
    k_vit_patchify<<<P, 256, 0, st>>>(pixels + (size_t)i * 3 * img * img, img, h->cfg.patch_size, h->Kp, h->ba_hi, h->ba_lo, i * Tp);
    DD_CHECK_LAUNCH();
  }
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = h->ba_hi, g.a_lo = h->ba_lo, g.M = M, g.S = h->Sp, g.W = h->wpatch, g.n_tiles = d / 16, g.bias = h->bpatch;
  g.out = h->bx, g.ldo = d, g.n_valid = d;
  RC(ddk_gemm(EPI_STORE, g, st));
  k_vit_embed_batch<<<M, 256, 0, st>>>(h->bx, h->cls, h->pos, d, T, Tp);
  DD_CHECK_LAUNCH();
  if (!h->no_pre_ln) {
    k_layernorm<<<M, 256, 0, st>>>(h->bx, d, h->prew, h->preb, h->cfg.ln_eps, h->bx, nullptr, nullptr, 0);
    DD_CHECK_LAUNCH();
  }
  for (auto& w : h->lw) {
    k_layernorm<<<M, 256, 0, st>>>(h->bx, d, w.ln1w, w.ln1b, h->cfg.ln_eps, nullptr, h->ba_hi, h->ba_lo, 0);
    DD_CHECK_LAUNCH();
    memset(&g, 0, sizeof(g));
    g.a_hi = h->ba_hi, g.a_lo = h->ba_lo, g.M = M, g.S = d / 32, g.W = w.wqkv, g.n_tiles = 3 * d / 16, g.bias = w.bqkv;
    g.qbuf = h->bq, g.kc = h->bkt, g.vc = h->bv, g.T_cap = Tp, g.vit_hidden = d, g.vit_head_dim = h->hd, g.vit_head_pad = h->hp;
    g.vit_qscale = 1.0f / sqrtf((float)h->hd);
    g.vit_img_rows = Tp, g.vit_T = T, g.vit_k_stride = img_kv, g.vit_v_stride = img_kv;
    RC(ddk_gemm(EPI_QKV_VIT, g, st));
    RC(ddk_attn_vit_mfma_batch(h->bq, h->btab, nb, Tp, T, Tp, h->H, h->ba_hi, h->ba_lo, st, h->hp));
    memset(&g, 0, sizeof(g));
    g.a_hi = h->ba_hi, g.a_lo = h->ba_lo, g.M = M, g.S = h->H * h->hp / 32, g.W = w.wo, g.n_tiles = d / 16, g.bias = w.bo;
    g.out = h->bx, g.ldo = d;
    RC(ddk_gemm(EPI_RESID, g, st));
    k_layernorm<<<M, 256, 0, st>>>(h->bx, d, w.ln2w, w.ln2b, h->cfg.ln_eps, nullptr, h->ba_hi, h->ba_lo, 0);
    DD_CHECK_LAUNCH();
    memset(&g, 0, sizeof(g));
    g.a_hi = h->ba_hi, g.a_lo = h->ba_lo, g.M = M, g.S = d / 32, g.W = w.wfc1, g.n_tiles = dff / 16, g.bias = w.bfc1;
    g.act = h->cfg.act, g.o_hi = h->bb_hi, g.o_lo = h->bb_lo, g.ld_planes = dff;
    RC(ddk_gemm(EPI_ACT, g, st));
    memset(&g, 0, sizeof(g));
    g.a_hi = h->bb_hi, g.a_lo = h->bb_lo, g.M = M, g.S = dff / 32, g.W = w.wfc2, g.n_tiles = d / 16, g.bias = w.bfc2;
    g.out = h->bx, g.ldo = d;
    RC(ddk_gemm(EPI_RESID, g, st));
  }
  if (!h->proj) {   // raw features of the selected layer, per image rows r0 .. r0 + nr - 1, optionally post-normed
    const int r0 = h->keep_cls ? 0 : 1, nr = h->keep_cls ? T : P;
    k_layernorm<<<nb * nr, 256, 0, st>>>(h->bx, d, h->post_ln ? h->postw : nullptr, h->post_ln ? h->postb : nullptr, h->cfg.ln_eps, out, nullptr,
                                         nullptr, r0, nr, Tp);
    DD_CHECK_LAUNCH();
    return DD_OK;
  }
  // projector on the P patch tokens of every image: gathered into nb * P contiguous rows = the layout of `out`
  const int Mp = nb * P;
  k_layernorm<<<Mp, 256, 0, st>>>(h->bx, d, nullptr, nullptr, 0.f, nullptr, h->ba_hi, h->ba_lo, 1, P, Tp);
  DD_CHECK_LAUNCH();
  memset(&g, 0, sizeof(g));
  g.a_hi = h->ba_hi, g.a_lo = h->ba_lo, g.M = Mp, g.S = d / 32, g.W = h->wp1, g.n_tiles = h->proj / 16, g.bias = h->bp1;
  g.act = 1, g.o_hi = h->bb_hi, g.o_lo = h->bb_lo, g.ld_planes = h->proj;
  RC(ddk_gemm(EPI_ACT, g, st));
  memset(&g, 0, sizeof(g));
  g.a_hi = h->bb_hi, g.a_lo = h->bb_lo, g.M = Mp, g.S = h->proj / 32, g.W = h->wp2, g.n_tiles = h->proj / 16, g.bias = h->bp2;
  g.out = out, g.ldo = h->proj, g.n_valid = h->proj;
  RC(ddk_gemm(EPI_STORE, g, st));
  return DD_OK;
}

extern "C" int dd_vit_forward(dd_vit* h, const float* pixels, int n_images, float* out, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && pixels && out && n_images >= 1, "dd_vit_forward: bad arguments");
  const int d = h->d, dff = h->dff, T = h->T, P = h->P, img = h->cfg.image_size;
  const int od = h->proj ? h->proj : d;
  if (n_images > 1 && ddk_prefill_mfma_enabled()) {      // several images: their tokens as one matrix, up to 16 at a time
    const int r_out = h->proj ? P : (h->keep_cls ? T : P);
    for (int i0 = 0; i0 < n_images; i0 += 16) {
      const int nb = n_images - i0 < 16 ? n_images - i0 : 16;
      RC(vit_forward_batch(h, pixels + (size_t)i0 * 3 * img * img, nb, out + (size_t)i0 * r_out * od, st));
    }
    return DD_OK;
  }
  for (int im = 0; im < n_images; ++im) {
    const float* px = pixels + (size_t)im * 3 * img * img;
    float* o = out + (size_t)im * P * od;
    k_vit_patchify<<<P, 256, 0, st>>>(px, img, h->cfg.patch_size, h->Kp, h->a_hi, h->a_lo);
    DD_CHECK_LAUNCH();
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a_hi = h->a_hi, g.a_lo = h->a_lo, g.M = T, g.S = h->Sp, g.W = h->wpatch, g.n_tiles = d / 16, g.bias = h->bpatch;   // bias zero unless loaded
    g.out = h->x, g.ldo = d, g.n_valid = d;                      // row 0 (class token) is overwritten next
    RC(ddk_gemm(EPI_STORE, g, st));
    k_vit_embed<<<T, 256, 0, st>>>(h->x, h->cls, h->pos, d);
    DD_CHECK_LAUNCH();
    if (!h->no_pre_ln) {
      k_layernorm<<<T, 256, 0, st>>>(h->x, d, h->prew, h->preb, h->cfg.ln_eps, h->x, nullptr, nullptr, 0);   // pre_layrnorm (CLIP)
      DD_CHECK_LAUNCH();
    }
    for (auto& w : h->lw) {
      k_layernorm<<<T, 256, 0, st>>>(h->x, d, w.ln1w, w.ln1b, h->cfg.ln_eps, nullptr, h->a_hi, h->a_lo, 0);
      DD_CHECK_LAUNCH();
      memset(&g, 0, sizeof(g));
      g.a_hi = h->a_hi, g.a_lo = h->a_lo, g.M = T, g.S = d / 32, g.W = w.wqkv, g.n_tiles = 3 * d / 16, g.bias = w.bqkv;
      g.qbuf = h->q, g.kc = h->kt, g.vc = h->v, g.T_cap = h->Tc, g.vit_hidden = d, g.vit_head_dim = h->hd, g.vit_head_pad = h->hp;
      g.vit_qscale = 1.0f / sqrtf((float)h->hd);
      RC(ddk_gemm(EPI_QKV_VIT, g, st));
      if (h->hp != 64 || ddk_prefill_mfma_enabled()) {
        RC(ddk_attn_vit_mfma(h->q, h->kt, h->v, T, h->Tc, h->H, h->a_hi, h->a_lo, st, h->hp));
      } else {
        k_attn_vit<<<dim3(h->H, (T + 3) / 4), 256, 0, st>>>(h->q, h->kt, h->v, T, h->Tc, d, h->a_hi, h->a_lo);
        DD_CHECK_LAUNCH();
      }
      memset(&g, 0, sizeof(g));
      g.a_hi = h->a_hi, g.a_lo = h->a_lo, g.M = T, g.S = h->H * h->hp / 32, g.W = w.wo, g.n_tiles = d / 16, g.bias = w.bo;
      g.out = h->x, g.ldo = d;
      RC(ddk_gemm(EPI_RESID, g, st));
      k_layernorm<<<T, 256, 0, st>>>(h->x, d, w.ln2w, w.ln2b, h->cfg.ln_eps, nullptr, h->a_hi, h->a_lo, 0);
      DD_CHECK_LAUNCH();
      memset(&g, 0, sizeof(g));
      g.a_hi = h->a_hi, g.a_lo = h->a_lo, g.M = T, g.S = d / 32, g.W = w.wfc1, g.n_tiles = dff / 16, g.bias = w.bfc1;
      g.act = h->cfg.act, g.o_hi = h->b_hi, g.o_lo = h->b_lo, g.ld_planes = dff;
      RC(ddk_gemm(EPI_ACT, g, st));
      memset(&g, 0, sizeof(g));
      g.a_hi = h->b_hi, g.a_lo = h->b_lo, g.M = T, g.S = dff / 32, g.W = w.wfc2, g.n_tiles = d / 16, g.bias = w.bfc2;
      g.out = h->x, g.ldo = d;
      RC(ddk_gemm(EPI_RESID, g, st));
    }
    if (!h->proj) {   // raw features of the selected layer: class token dropped (CLIP feature layer) or kept, optionally post-normed
      const int r0 = h->keep_cls ? 0 : 1, nr = h->keep_cls ? T : P;
      float* oo = out + (size_t)im * nr * d;
      if (h->post_ln) {
        k_layernorm<<<nr, 256, 0, st>>>(h->x, d, h->postw, h->postb, h->cfg.ln_eps, oo, nullptr, nullptr, r0);
        DD_CHECK_LAUNCH();
      } else {
        DD_HIP(hipMemcpyAsync(oo, h->x + (size_t)r0 * d, (size_t)nr * d * 4, hipMemcpyDeviceToDevice, st));
      }
      continue;
    }
    // projector: linear_1 -> GELU(erf) -> linear_2 on rows 1..P  (LlavaMultiModalProjector)
    k_layernorm<<<P, 256, 0, st>>>(h->x, d, nullptr, nullptr, 0.f, nullptr, h->a_hi, h->a_lo, 1);
    DD_CHECK_LAUNCH();
    memset(&g, 0, sizeof(g));
    g.a_hi = h->a_hi, g.a_lo = h->a_lo, g.M = P, g.S = d / 32, g.W = h->wp1, g.n_tiles = h->proj / 16, g.bias = h->bp1;
    g.act = 1, g.o_hi = h->b_hi, g.o_lo = h->b_lo, g.ld_planes = h->proj;
    RC(ddk_gemm(EPI_ACT, g, st));
    memset(&g, 0, sizeof(g));
    g.a_hi = h->b_hi, g.a_lo = h->b_lo, g.M = P, g.S = h->proj / 32, g.W = h->wp2, g.n_tiles = h->proj / 16, g.bias = h->bp2;
    g.out = o, g.ldo = h->proj, g.n_valid = h->proj;
    RC(ddk_gemm(EPI_STORE, g, st));
  }
  return DD_OK;
}

// =======================================================================================================================
// InstructBLIP Q-Former + language projection (reference models/instructblip.py:613-633 calls HF's
// InstructBlipQFormerModel and nn.Linear).  BERT-style post-LayerNorm encoder over S = Q + n rows (query tokens, then the
// instruction tokens): self-attention over all rows; on layers l % freq == 0 the query rows cross-attend to the vision
// tokens; the query rows and the instruction rows run separate GELU feed-forward blocks.  Every matrix product runs on
// k_gemm (bf16 weights, fp32 activations as hi/lo planes), attention on the matrix-core kernel shared with the towers,
// LayerNorm / residual stream in fp32.
// =======================================================================================================================
struct QfLayer {
  u32x4_t *wqkv = nullptr, *wo = nullptr, *cwq = nullptr, *cwkv = nullptr, *cwo = nullptr, *wq1 = nullptr, *wq2 = nullptr, *wt1 = nullptr,
          *wt2 = nullptr;
  float *bqkv = nullptr, *bo = nullptr, *lnw = nullptr, *lnb = nullptr;
  float *cbq = nullptr, *cbkv = nullptr, *cbo = nullptr, *clnw = nullptr, *clnb = nullptr;
  float *bq1 = nullptr, *bq2 = nullptr, *qlnw = nullptr, *qlnb = nullptr, *bt1 = nullptr, *bt2 = nullptr, *tlnw = nullptr, *tlnb = nullptr;
  bool cross = false;
};

struct dd_qformer {
  dd_qformer_config cfg;
  int d, H, dff, de, Q, Sc, Ec, proj;     // Sc / Ec: row capacities (multiples of 64) of the sequence and of the vision tokens
  std::vector<void*> allocs;
  std::vector<QfLayer> lw;
  float *word = nullptr, *posemb = nullptr, *elnw = nullptr, *elnb = nullptr, *qtok = nullptr, *pb = nullptr;
  u32x4_t* pw = nullptr;
  // scratch
  float *x = nullptr, *q = nullptr, *kt = nullptr, *v = nullptr, *ekt = nullptr, *ev = nullptr;
  uint16_t *a_hi = nullptr, *a_lo = nullptr, *b_hi = nullptr, *b_lo = nullptr, *e_hi = nullptr, *e_lo = nullptr;
};

template <typename T>
static int qalloc(dd_qformer* h, T** p, size_t n) {
  void* q = nullptr;
  size_t b = n * sizeof(T);
  if (b == 0) b = 16;
  if (hipMalloc(&q, b) != hipSuccess) {
    dd_set_error("dd_qformer: hipMalloc(%zu) failed", b);
    return DD_ENOMEM;
  }
  (void)hipMemset(q, 0, b);
  h->allocs.push_back(q);
  *p = (T*)q;
  return DD_OK;
}
#define QA(ptr, n)                                   \
  do {                                               \
    int rc__ = qalloc(h, &(ptr), (size_t)(n));       \
    if (rc__ != DD_OK) {                             \
      dd_qformer_destroy(h);                         \
      return rc__;                                   \
    }                                                \
  } while (0)

extern "C" int dd_qformer_destroy(dd_qformer* h) {
  if (!h) return DD_OK;
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return DD_OK;
}

extern "C" int dd_qformer_create(const dd_qformer_config* c, dd_qformer** out) {
  DD_REQUIRE(c && out, "dd_qformer_create: null argument");
  DD_REQUIRE(c->hidden_size % 64 == 0 && c->intermediate_size % 64 == 0 && c->encoder_hidden_size % 64 == 0 && c->proj_dim % 16 == 0,
             "dd_qformer_create: hidden / intermediate / encoder widths must be multiples of 64 (projection: of 16)");
  DD_REQUIRE(c->num_heads > 0 && c->hidden_size == c->num_heads * 64, "dd_qformer_create: head_dim must be 64, got %d / %d", c->hidden_size,
             c->num_heads);
  DD_REQUIRE(c->num_layers >= 1 && c->cross_attention_frequency >= 1 && c->num_query_tokens >= 1, "dd_qformer_create: layers / frequency / queries");
  DD_REQUIRE(c->max_text_tokens >= 0 && c->max_text_tokens <= c->max_position_embeddings, "dd_qformer_create: max_text_tokens %d exceeds the position table (%d)",
             c->max_text_tokens, c->max_position_embeddings);
  DD_REQUIRE(c->max_encoder_tokens >= 1, "dd_qformer_create: max_encoder_tokens");
  dd_qformer* h = new dd_qformer();
  h->cfg = *c;
  const int d = h->d = c->hidden_size, dff = h->dff = c->intermediate_size, de = h->de = c->encoder_hidden_size;
  h->H = c->num_heads, h->Q = c->num_query_tokens, h->proj = c->proj_dim;
  h->Sc = (c->num_query_tokens + c->max_text_tokens + 63) / 64 * 64;
  h->Ec = (c->max_encoder_tokens + 63) / 64 * 64;
  h->lw.resize(c->num_layers);
  for (int l = 0; l < c->num_layers; ++l) {
    QfLayer& w = h->lw[l];
    w.cross = l % c->cross_attention_frequency == 0;
    QA(w.wqkv, (size_t)(3 * d / 16) * (d / 32) * 64); QA(w.wo, (size_t)(d / 16) * (d / 32) * 64);
    QA(w.bqkv, 3 * d); QA(w.bo, d); QA(w.lnw, d); QA(w.lnb, d);
    if (w.cross) {
      QA(w.cwq, (size_t)(d / 16) * (d / 32) * 64); QA(w.cwkv, (size_t)(2 * d / 16) * (de / 32) * 64); QA(w.cwo, (size_t)(d / 16) * (d / 32) * 64);
      QA(w.cbq, d); QA(w.cbkv, 2 * d); QA(w.cbo, d); QA(w.clnw, d); QA(w.clnb, d);
    }
    QA(w.wq1, (size_t)(dff / 16) * (d / 32) * 64); QA(w.wq2, (size_t)(d / 16) * (dff / 32) * 64);
    QA(w.wt1, (size_t)(dff / 16) * (d / 32) * 64); QA(w.wt2, (size_t)(d / 16) * (dff / 32) * 64);
    QA(w.bq1, dff); QA(w.bq2, d); QA(w.qlnw, d); QA(w.qlnb, d); QA(w.bt1, dff); QA(w.bt2, d); QA(w.tlnw, d); QA(w.tlnb, d);
  }
  QA(h->word, (size_t)c->vocab_size * d); QA(h->posemb, (size_t)c->max_position_embeddings * d);
  QA(h->elnw, d); QA(h->elnb, d); QA(h->qtok, (size_t)h->Q * d);
  QA(h->pw, (size_t)(h->proj / 16) * (d / 32) * 64); QA(h->pb, h->proj);
  QA(h->x, (size_t)h->Sc * d); QA(h->q, (size_t)h->Sc * d); QA(h->kt, (size_t)h->Sc * d); QA(h->v, (size_t)h->Sc * d);
  QA(h->ekt, (size_t)h->Ec * d); QA(h->ev, (size_t)h->Ec * d);
  QA(h->a_hi, (size_t)h->Sc * d); QA(h->a_lo, (size_t)h->Sc * d);
  QA(h->b_hi, (size_t)h->Sc * dff); QA(h->b_lo, (size_t)h->Sc * dff);
  QA(h->e_hi, (size_t)h->Ec * de); QA(h->e_lo, (size_t)h->Ec * de);
  *out = h;
  return DD_OK;
}

extern "C" int dd_qformer_load_tensor(dd_qformer* h, int id, int layer, const uint16_t* src, int rows, int cols, int on_device) {
  DD_REQUIRE(h && src, "dd_qformer_load_tensor: null argument");
  const bool per_layer = id >= DD_QF_SA_WQ;
  DD_REQUIRE((id >= DD_QF_WORD_EMB && id <= DD_QF_PROJ_B) || (id >= DD_QF_SA_WQ && id <= DD_QF_FFT_LN_B), "dd_qformer_load_tensor: unknown tensor id %d", id);
  DD_REQUIRE(!per_layer || (layer >= 0 && layer < (int)h->lw.size()), "dd_qformer_load_tensor: layer %d out of range", layer);
  const int d = h->d, dff = h->dff, de = h->de;
  QfLayer* w = per_layer ? &h->lw[layer] : nullptr;
  DD_REQUIRE(!(per_layer && id >= DD_QF_CA_WQ && id <= DD_QF_CA_LN_B) || w->cross, "dd_qformer_load_tensor: layer %d has no cross-attention", layer);
  size_t n = (size_t)rows * cols;
  const uint16_t* dev = src;
  uint16_t* staging = nullptr;
  if (!on_device) {
    DD_HIP(hipMalloc((void**)&staging, n * 2));
    if (hipMemcpy(staging, src, n * 2, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(staging);
      dd_set_error("dd_qformer_load_tensor: H2D copy failed");
      return DD_EHIP;
    }
    dev = staging;
  }
  int rc = DD_OK;
  auto vec = [&](float* dst, size_t len) -> int {
    if (len != n) {
      dd_set_error("dd_qformer_load_tensor: tensor %d expects %zu values, got %zu", id, len, n);
      return DD_EINVAL;
    }
    return ddk_bf16_to_f32(dev, dst, (int)len, nullptr);
  };
  auto mat = [&](u32x4_t* dst, int er, int ec, int tile0) -> int {
    if (rows != er || cols != ec) {
      dd_set_error("dd_qformer_load_tensor: tensor %d expects %d x %d, got %d x %d", id, er, ec, rows, cols);
      return DD_EINVAL;
    }
    return ddk_pack_weight(dev, rows, cols, dst, tile0, 1, PACK_PLAIN, er / 16, nullptr);
  };
  switch (id) {
    case DD_QF_WORD_EMB: rc = vec(h->word, (size_t)h->cfg.vocab_size * d); break;
    case DD_QF_POS_EMB: rc = vec(h->posemb, (size_t)h->cfg.max_position_embeddings * d); break;
    case DD_QF_EMB_LN_W: rc = vec(h->elnw, d); break;
    case DD_QF_EMB_LN_B: rc = vec(h->elnb, d); break;
    case DD_QF_QUERY_TOKENS: rc = vec(h->qtok, (size_t)h->Q * d); break;
    case DD_QF_PROJ_W: rc = mat(h->pw, h->proj, d, 0); break;
    case DD_QF_PROJ_B: rc = vec(h->pb, h->proj); break;
    case DD_QF_SA_WQ: rc = mat(w->wqkv, d, d, 0); break;
    case DD_QF_SA_WK: rc = mat(w->wqkv, d, d, d / 16); break;
    case DD_QF_SA_WV: rc = mat(w->wqkv, d, d, 2 * d / 16); break;
    case DD_QF_SA_BQ: rc = vec(w->bqkv, d); break;
    case DD_QF_SA_BK: rc = vec(w->bqkv + d, d); break;
    case DD_QF_SA_BV: rc = vec(w->bqkv + 2 * d, d); break;
    case DD_QF_SA_WO: rc = mat(w->wo, d, d, 0); break;
    case DD_QF_SA_BO: rc = vec(w->bo, d); break;
    case DD_QF_SA_LN_W: rc = vec(w->lnw, d); break;
    case DD_QF_SA_LN_B: rc = vec(w->lnb, d); break;
    case DD_QF_CA_WQ: rc = mat(w->cwq, d, d, 0); break;
    case DD_QF_CA_BQ: rc = vec(w->cbq, d); break;
    case DD_QF_CA_WK: rc = mat(w->cwkv, d, de, 0); break;
    case DD_QF_CA_BK: rc = vec(w->cbkv, d); break;
    case DD_QF_CA_WV: rc = mat(w->cwkv, d, de, d / 16); break;
    case DD_QF_CA_BV: rc = vec(w->cbkv + d, d); break;
    case DD_QF_CA_WO: rc = mat(w->cwo, d, d, 0); break;
    case DD_QF_CA_BO: rc = vec(w->cbo, d); break;
    case DD_QF_CA_LN_W: rc = vec(w->clnw, d); break;
    case DD_QF_CA_LN_B: rc = vec(w->clnb, d); break;
    case DD_QF_FFQ_W1: rc = mat(w->wq1, dff, d, 0); break;
    case DD_QF_FFQ_B1: rc = vec(w->bq1, dff); break;
    case DD_QF_FFQ_W2: rc = mat(w->wq2, d, dff, 0); break;
    case DD_QF_FFQ_B2: rc = vec(w->bq2, d); break;
    case DD_QF_FFQ_LN_W: rc = vec(w->qlnw, d); break;
    case DD_QF_FFQ_LN_B: rc = vec(w->qlnb, d); break;
    case DD_QF_FFT_W1: rc = mat(w->wt1, dff, d, 0); break;
    case DD_QF_FFT_B1: rc = vec(w->bt1, dff); break;
    case DD_QF_FFT_W2: rc = mat(w->wt2, d, dff, 0); break;
    case DD_QF_FFT_B2: rc = vec(w->bt2, d); break;
    case DD_QF_FFT_LN_W: rc = vec(w->tlnw, d); break;
    case DD_QF_FFT_LN_B: rc = vec(w->tlnb, d); break;
    default: rc = DD_EINVAL; dd_set_error("dd_qformer_load_tensor: unknown tensor id %d", id);
  }
  hipError_t e = hipDeviceSynchronize();
  if (staging) (void)hipFree(staging);
  if (rc != DD_OK) return rc;
  DD_HIP(e);
  return DD_OK;
}

// rows 0..Q-1 = query tokens; row Q + i = word_embeddings[ids[i]] + position_embeddings[i]   (InstructBlipQFormerEmbeddings,
// before its LayerNorm).  An id outside the table raises the flag (checked by the caller through the returned rows: NaN).
__global__ __launch_bounds__(256) void k_qf_embed(float* __restrict__ x, const float* __restrict__ qtok, const float* __restrict__ word,
                                                  const float* __restrict__ pos, const int32_t* __restrict__ ids, int Q, int d, int vocab) {
  const int row = blockIdx.x;
  for (int i = threadIdx.x; i < d; i += 256) {
    float v;
    if (row < Q) {
      v = qtok[(size_t)row * d + i];
    } else {
      int id = ids[row - Q];
      v = (id >= 0 && id < vocab) ? word[(size_t)id * d + i] + pos[(size_t)(row - Q) * d + i] : __builtin_nanf("");
    }
    x[(size_t)row * d + i] = v;
  }
}

extern "C" int dd_qformer_forward(dd_qformer* h, const int32_t* text_ids, int n_text, const float* enc, int n_enc, float* out,
                                  float* hidden_out, void* stream_) {
  hipStream_t st = (hipStream_t)stream_;
  DD_REQUIRE(h && enc && out, "dd_qformer_forward: null argument");
  DD_REQUIRE(n_text >= 0 && n_text <= h->cfg.max_text_tokens, "dd_qformer_forward: %d instruction tokens, capacity %d", n_text, h->cfg.max_text_tokens);
  DD_REQUIRE(n_text == 0 || text_ids, "dd_qformer_forward: text ids missing");
  DD_REQUIRE(n_enc >= 1 && n_enc <= h->cfg.max_encoder_tokens, "dd_qformer_forward: %d vision tokens, capacity %d", n_enc, h->cfg.max_encoder_tokens);
  const int d = h->d, dff = h->dff, de = h->de, Q = h->Q, S = Q + n_text;
  const float eps = h->cfg.ln_eps;
  GemmArgs g;
  // embeddings + LayerNorm
  k_qf_embed<<<S, 256, 0, st>>>(h->x, h->qtok, h->word, h->posemb, text_ids, Q, d, h->cfg.vocab_size);
  DD_CHECK_LAUNCH();
  k_layernorm<<<S, 256, 0, st>>>(h->x, d, h->elnw, h->elnb, eps, h->x, nullptr, nullptr, 0);
  DD_CHECK_LAUNCH();
  // vision tokens as operand planes, once for all cross-attention layers
  k_layernorm<<<n_enc, 256, 0, st>>>(enc, de, nullptr, nullptr, 0.f, nullptr, h->e_hi, h->e_lo, 0);
  DD_CHECK_LAUNCH();
  // x[r0 .. r0+M) += planes(b) . W2^T + b2, then LayerNorm in place (BERT output block)
  auto out_block = [&](uint16_t* p_hi, uint16_t* p_lo, int K, u32x4_t* W, float* bias, float* lnw, float* lnb, int r0, int M) -> int {
    GemmArgs o;
    memset(&o, 0, sizeof(o));
    o.a_hi = p_hi, o.a_lo = p_lo, o.M = M, o.S = K / 32, o.W = W, o.n_tiles = d / 16, o.bias = bias;
    o.out = h->x + (size_t)r0 * d, o.ldo = d;
    RC(ddk_gemm(EPI_RESID, o, st));
    k_layernorm<<<M, 256, 0, st>>>(h->x + (size_t)r0 * d, d, lnw, lnb, eps, h->x + (size_t)r0 * d, nullptr, nullptr, 0);
    DD_CHECK_LAUNCH();
    return DD_OK;
  };
  auto ffn = [&](u32x4_t* W1, float* b1, u32x4_t* W2, float* b2, float* lnw, float* lnb, int r0, int M) -> int {
    k_layernorm<<<M, 256, 0, st>>>(h->x, d, nullptr, nullptr, 0.f, nullptr, h->a_hi, h->a_lo, r0);   // plain split of rows r0..
    DD_CHECK_LAUNCH();
    GemmArgs f;
    memset(&f, 0, sizeof(f));
    f.a_hi = h->a_hi, f.a_lo = h->a_lo, f.M = M, f.S = d / 32, f.W = W1, f.n_tiles = dff / 16, f.bias = b1;
    f.act = 1, f.o_hi = h->b_hi, f.o_lo = h->b_lo, f.ld_planes = dff;                                  // GELU (erf)
    RC(ddk_gemm(EPI_ACT, f, st));
    return out_block(h->b_hi, h->b_lo, dff, W2, b2, lnw, lnb, r0, M);
  };
  for (auto& w : h->lw) {
    // self-attention over all S rows
    k_layernorm<<<S, 256, 0, st>>>(h->x, d, nullptr, nullptr, 0.f, nullptr, h->a_hi, h->a_lo, 0);
    DD_CHECK_LAUNCH();
    memset(&g, 0, sizeof(g));
    g.a_hi = h->a_hi, g.a_lo = h->a_lo, g.M = S, g.S = d / 32, g.W = w.wqkv, g.n_tiles = 3 * d / 16, g.bias = w.bqkv;
    g.qbuf = h->q, g.kc = h->kt, g.vc = h->v, g.T_cap = h->Sc, g.vit_hidden = d, g.vit_head_dim = 64, g.vit_qscale = 0.125f;
    RC(ddk_gemm(EPI_QKV_VIT, g, st));
    RC(ddk_attn_vit_mfma(h->q, h->kt, h->v, S, h->Sc, h->H, h->a_hi, h->a_lo, st, 64));
    RC(out_block(h->a_hi, h->a_lo, d, w.wo, w.bo, w.lnw, w.lnb, 0, S));
    if (w.cross) {   // query rows against the vision tokens
      k_layernorm<<<Q, 256, 0, st>>>(h->x, d, nullptr, nullptr, 0.f, nullptr, h->a_hi, h->a_lo, 0);
      DD_CHECK_LAUNCH();
      memset(&g, 0, sizeof(g));
      g.a_hi = h->a_hi, g.a_lo = h->a_lo, g.M = Q, g.S = d / 32, g.W = w.cwq, g.n_tiles = d / 16, g.bias = w.cbq;
      g.qbuf = h->q, g.kc = h->ekt, g.vc = h->ev, g.T_cap = h->Ec, g.vit_hidden = d, g.vit_head_dim = 64, g.vit_qscale = 0.125f;
      RC(ddk_gemm(EPI_QKV_VIT, g, st));
      memset(&g, 0, sizeof(g));
      g.a_hi = h->e_hi, g.a_lo = h->e_lo, g.M = n_enc, g.S = de / 32, g.W = w.cwkv, g.n_tiles = 2 * d / 16, g.bias = w.cbkv;
      g.qbuf = h->q, g.kc = h->ekt, g.vc = h->ev, g.T_cap = h->Ec, g.vit_hidden = d, g.vit_head_dim = 64, g.vit_qscale = 1.0f;
      g.vit_col0 = d;                                                           // [k | v] columns of a fused [q | k | v]
      RC(ddk_gemm(EPI_QKV_VIT, g, st));
      RC(ddk_attn_vit_mfma(h->q, h->ekt, h->ev, Q, h->Ec, h->H, h->a_hi, h->a_lo, st, 64, n_enc));
      RC(out_block(h->a_hi, h->a_lo, d, w.cwo, w.cbo, w.clnw, w.clnb, 0, Q));
    }
    RC(ffn(w.wq1, w.bq1, w.wq2, w.bq2, w.qlnw, w.qlnb, 0, Q));
    if (n_text > 0) RC(ffn(w.wt1, w.bt1, w.wt2, w.bt2, w.tlnw, w.tlnb, Q, n_text));
  }
  if (hidden_out) DD_HIP(hipMemcpyAsync(hidden_out, h->x, (size_t)S * d * 4, hipMemcpyDeviceToDevice, st));
  // language_projection on the query rows
  k_layernorm<<<Q, 256, 0, st>>>(h->x, d, nullptr, nullptr, 0.f, nullptr, h->a_hi, h->a_lo, 0);
  DD_CHECK_LAUNCH();
  memset(&g, 0, sizeof(g));
  g.a_hi = h->a_hi, g.a_lo = h->a_lo, g.M = Q, g.S = d / 32, g.W = h->pw, g.n_tiles = h->proj / 16, g.bias = h->pb;
  g.out = out, g.ldo = h->proj, g.n_valid = h->proj;
  RC(ddk_gemm(EPI_STORE, g, st));
  return DD_OK;
}
