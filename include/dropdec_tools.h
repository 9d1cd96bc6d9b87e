/* Measurement hooks and experiment knobs of libdropdec_tools.so — NOT part of the drop-in boundary (include/dropdec.h).
 *
 * libdropdec_tools.so = the objects of libdropdec.so + csrc/dd_tools.hip.  bench.py's roofline leg and the scripts under
 * tools/ load it instead of the product library; handles of one library must not be passed to the other (each has its own
 * state).  Nothing here replaces a reference interface: the reference has no native code and no timing hooks. */
#ifndef DROPDEC_TOOLS_H
#define DROPDEC_TOOLS_H
#include "dropdec.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Time the packed member sweep alone with HIP events on `stream`: runs `iters` sweeps of
 * `nb` rows at the current length and returns the mean milliseconds (bench.py roofline leg). */
int dd_lm_time_sweep(dd_lm* h, int nb, int iters, float* mean_ms_out, void* stream);

/* Time one decode GEMV kind in isolation (HIP events on `stream`), cycling over the layers' weights so every
 * launch streams bytes that are not cache-resident. which: 0 qkv, 1 o_proj, 2 gate/up, 3 down_proj; nb = rows (1..8, or
 * 16 / 32 / 64 = the two- / four- / eight-plane kernels of the lanes path: k_gemv_slices + k_gemv_finish, or k_gemv_groups).
 * which + 8: the slice-resident path's STREAMING kernel alone (without its finishing kernel) — the kernel
 * whose duration rocprofv3's kernel trace reports under the name dd_tools_last_gemv_kernel() returns afterwards.
 * bytes_per_launch_out = algorithmic (weight) bytes of one launch. */
int dd_lm_time_gemv(dd_lm* h, int which, int nb, int iters, float* mean_ms_out, double* bytes_per_launch_out,
                    void* stream);

/* Name, as a kernel trace prints it (template arguments included), of the decode-GEMV streaming kernel the calling
 * thread's last GEMV launch used. */
const char* dd_tools_last_gemv_kernel(void);

/* Streaming READ bandwidth (GB/s) this device delivers over buf_dev[bytes] (bytes >= 1 MiB; use a buffer much larger than
 * the 256 MiB Infinity Cache), HIP events on `stream`. */
int dd_hbm_read_bench(const void* buf_dev, size_t bytes, int iters, int n_blocks, float* gbs_out, void* stream);

/* A/B switches of kernel variants (every setting produces the same bits); keys as tools/ and DESIGN.md quote them:
 * 0 GEMV weight tiles in flight per wave, 4 ring / batch request order of the 8-row GEMV, 9 sequences per member sweep of
 * dd_lm_group_step (1, 2, 4, 8), 10 workgroups per group of 8 members in the grouped decode attention, 12 prefill attention
 * on the matrix cores, 17 / 18 / 19 workgroups per K slice of the 64-row qkv / o_proj / gate-up GEMV, 21 key tiles per
 * workgroup of the fp16-cache decode attention, 22 all-tiles form of that attention, 23 member sweeps of a group step that
 * run concurrently (1..4), 24 four-columns-per-thread finishing kernel of the slice GEMVs (bit mask over the epilogues
 * EPI_STORE / RESID / SILU / QKV, default 15; 0: the one-column kernel it replaced), 26 rider form of the group step (default 1;
 * 0: always the classic form: one fused un-masked sweep + the member sweeps), 27 the riding rows' attention inside the members'
 * launches (default 1), 28 branches of the rider form (1..4, default 4), 29 weight requests in flight per wave of the nine-plane
 * qkv / gate-up kernels (4 or 8, default 4), 30 half planes for K <= 4 (two sequences per operand plane, default 1), 31 half planes
 * before the rider form where both apply and the line-up is not whole groups of fourteen (default 1), 33 the rider form's rings in stages with
 * the masks of the groups whose rows rode sampled between the stages on the caller's stream (default 1; 0: on the branches), 34 the lanes
 * mask sampler in one of the 1,024-thread forms this library keeps (csrc/dd_sampler_block.h; default 0 = the product's one-wave kernel; 1 = round 3's
 * form with a private-scratch copy of its parameters, 616 bytes per lane; 2 = the checking form, dd_tools_sampler_dbg_attach; 3 = round 4's kernel).
 * Round 4: 36 default-policy instead of non-temporal weight loads in the slice GEMVs (default 0), 38 the GQA decode attention with all q heads
 * of a kv head in one workgroup (default 0: two heads per workgroup), 45 prefill RMSNorm + split with sixteen rows per workgroup (default 1;
 * 0: one row per workgroup), 46 prefill attention over the fp16 cache with operand-staged K / V tiles and 1 or 2 query blocks per wave
 * (default 1; 0: the fp32-staged kernel), 47 the MHA rider sweeps' attention tile pass with two register sets (the round-3 form, 144 VGPRs;
 * default 0: one set, 94 VGPRs, so that it shares CUs with the other branches' slice GEMVs), 48 the dynamic-LDS request of the 1,024-thread sampler forms (key 34 > 0): 1 = 156 KiB, round 4's
 * fence (default); 0 = the 76 KiB they use — the unit reproducer of DESIGN.md 3e needs that; n = n KiB (round 5's request sweep).  Round 5: 49 the
 * progressive stage-in of the operand planes in the whole-slice GEMV kernels at two and four planes (default 1; 0: blocking, as in rounds 2-4); 36
 * also takes timing-only bits: 2 = the slice kernels skip the stage-in, 4 = the slice-pair kernels write no partial sums (results garbage), 8 = the
 * slice-pair kernels write a tile pair's partial sums in mid-stream (as until round 5's last day; same bits); 50 the rows' rstd of a slice GEMV in a
 * workgroup of its own behind the streaming ones (default 1; 0: workgroup 0 computes it before its own weight stream; same bits).  Keys of the determinism bisect (all default to the product's behaviour): 37 fp32-cache
 * engines fork their member sweeps (-1: one branch as in round 3), 39 extra dynamic LDS bytes requested by the fp32-cache attention tile pass
 * (so that it cannot share a CU with a slice GEMV), 40 the branches' streams on disjoint CU masks, 41 CU-mask only the attention launches,
 * 42 bit mask of kernel families launched on the UNMASKED stream while 40 is on (1 embed, 2 GEMVs, 4 attention, 8 finishing kernels),
 * 43 the fp32-cache tile pass with scalar instead of packed FP32 multiply-adds in its P.V step (tools library only: the product library has
 * no packed FP32 at all), 52 (round 6) the nine-plane fp8 slice kernel's operand fragments through a ring read five ahead of their MFMAs
 * (default 1; 0: where the compiler requests them; same bits), 53 (round 6, TIMING ONLY: sums in an order no other width produces) four K slices
 * per workgroup in the nine-plane qkv / gate-up kernels (1: 96 / 86 workgroups per quad, 2: gate/up on 128), 54 (round 6) the 8-row GEMV that keeps
 * the rows' operand in registers and walks several tile groups per workgroup (k_gemv_loop: default 1 = where the tile count is a multiple of 256;
 * 0: never; n > 1: n workgroups for every K = 4096 matrix; same bits).  Keys of
 * dd_set_tuning are forwarded.  Every call starts a new epoch of the step-graph
 * keys: a step captured under other settings is never replayed. */
int dd_tools_set_tuning(int key, int value);

/* Determinism diagnostics (tools/stress_lanes.py).
 * dd_tools_trace_attach: from now on every group step appends one 32-int record per step to buf_dev[cap_steps][32] for this
 * sequence (tokens so far, un-masked argmax, keep-set size, hash of the drop bits, n_drop[8], member argmax ids[8], winner, token,
 * mt19937 read index, hash of the un-masked logits); buf_dev = NULL detaches.  rng: the sequence's generator (may be NULL).
 * dd_tools_lds_poison: `launches` grids of `wgs` workgroups that fill lds_bytes of LDS with a NaN pattern (beside a step: a kernel
 * that reads LDS it did not write turns it into a token change).
 * dd_tools_scratch_probe: `launches` grids of `wgs` 1,024-thread workgroups with 616 bytes of private scratch per lane that write a
 * pattern, linger for `spin` barrier rounds and verify it; errors_dev[0] += mismatching words. */
int dd_tools_trace_attach(dd_lm* h, int32_t* buf_dev, int cap_steps, dd_rng* rng);
/* The group step's mask sampler on its own (keep sets from argmax / top-k ids + the K masks of n <= 32 sequences, one workgroup per
 * sequence, each from its own generator): arrays of n device pointers. */
int dd_tools_sample_masks_lanes(int n, const float* const* epi, const int32_t* L, uint8_t* const* keep, const int32_t* const* argmax,
                                const int32_t* const* topk, dd_rng* const* rngs, uint8_t* const* drop, int32_t* const* n_drop,
                                uint8_t* const* drop_bits, int k_top, const double* mprobs, int K, int mode, void* stream);
int dd_tools_lds_poison(int launches, int wgs, int lds_bytes, void* stream);
/* Per-stage checksums of every multi-group sweep enqueued from now on (eager launches): trace_dev [cap_sweeps][n_layers][8] uint32, zeroed by the
 * caller; stage 0 embed, 1 q rows, 2 new K rows, 3 attention output, 4 o_proj, 5 gate/up, 6 down (rows), 7 down (next operand).  NULL: off. */
int dd_tools_sweep_trace(uint32_t* trace_dev, int cap_sweeps);
/* With dd_tools_sweep_trace: per-workgroup checksums inside the fp32-cache attention tile pass, attn_dev [cap_sweeps][n_layers][stride_words]
 * (8 words per workgroup: K registers, V registers, q rows read from LDS, scores read, p written, p read, outputs read, 0).  NULL: off. */
int dd_tools_attn_trace(uint32_t* attn_dev, size_t stride_words);
int dd_tools_scratch_probe(int launches, int wgs, int spin, unsigned int* errors_dev, void* stream);
/* The LDS exchange pattern of the fp32-cache attention tile pass (256 threads, 30,720 bytes of dynamic LDS, rows written by one wave and read by
 * the others across a barrier) with verifiable values: errors_dev[0] += mismatching words. */
int dd_tools_lds_barrier_probe(int launches, int wgs, int rounds, unsigned int* errors_dev, void* stream);
/* LDS overlap probe: a grid of 512-thread workgroups holding a verifiable pattern in lds_a bytes of dynamic LDS on stream_a while launches_b grids
 * of 256-thread workgroups do the same with lds_b bytes on stream_b; errors_dev[0] / [1] += corrupted words seen by A / B workgroups, [2] / [3] = lowest /
 * highest corrupted word offset in a B workgroup (initialise to 0xFFFFFFFF / 0). */
/* Register / load probes shaped like the fp32-cache attention tile pass (256 threads, ~120 VGPRs, 30,720 bytes of dynamic LDS): kind 0 holds a pattern in
 * 96 registers per lane across `hold` sleep + barrier rounds; kind 1 keeps 16 outstanding 16-byte global loads per lane from a 1 GiB buffer of known contents.
 * errors_dev[0] += mismatching words. */
/* v_pk_fma_f32 chains against the same multiply-adds as scalar v_fma_f32 on identical operands (256-thread workgroups, 30,720 bytes of dynamic LDS);
 * errors_dev[0] += lanes whose packed and scalar results differ. */
/* The P.V step of the fp32-cache attention tile pass as the compiler emits it (v_pk_fma_f32 fed by ds_read_b128) next to the same sums as scalar
 * v_fma_f32, compared bit for bit; errors_dev[0] += (lane, row) results that differ. */
int dd_tools_pv_probe(int launches, int wgs, int iters, unsigned int* errors_dev, void* stream);
/* Packed sums of eight 16-byte GLOBAL loads per thread (the slice GEMVs' finishing kernel) next to the same sums as scalar v_add_f32. */
int dd_tools_pkadd_gload_probe(int launches, int wgs, int iters, unsigned int* errors_dev, void* stream);
/* The mask sampler's co-residency fault shrunk to its victim phase: `wgs` workgroups of 1,024 threads with `lds_bytes` of dynamic
 * LDS (77,856 = the sampler's own request before its padding .. 159,744) regenerate an mt19937 block `iters` times the way the sampler does and
 * check every regeneration against an in-order recomputation by one wave.  out_dev[0] += differing words; out_dev[1..7] = the first difference
 * (workgroup, iteration, word index, got, want, neighbours).  tools/sampler_repro.py twist_probe runs it beside a group taking rider steps. */
int dd_tools_twist_probe(int launches, int wgs, int iters, int lds_bytes, unsigned int* out_dev, void* stream);
/* s_barrier looked at directly (round 5): `wgs` workgroups of `threads` (256 / 512 / 1024) threads with lds_bytes of dynamic LDS; per iteration every
 * thread stores the iteration number to its word of a block at the sampler's generator offset, barrier, reads the other waves' words, barrier, then
 * idles a wave-dependent while.  out_dev[0] += words that were not this iteration's, [1] += one iteration old, [2] += of a later iteration,
 * [3..7] = first event (workgroup, iteration, reader wave, writer wave, value). */
int dd_tools_barrier_probe(int launches, int wgs, int threads, int iters, int lds_bytes, unsigned int* out_dev, void* stream);
int dd_tools_pk_probe(int launches, int wgs, int iters, unsigned int* errors_dev, void* stream);
/* The packed-FP32 fault read as a write-after-read hazard (round 5): the P.V step with its registers fixed by hand — variant 0: the re-load
 * (ds_read_b128) overwrites the operand registers right behind the v_pk_fma_f32 that read them (the compiler's shape); 1: sixteen wait states
 * between; 2: the re-load goes to registers no packed op of the last twelve read.  errors_dev[0..2] += (lane, row) results of each variant
 * that differ from scalar v_fma_f32 sums of the same operands. */
int dd_tools_pk_war_probe(int launches, int wgs, int iters, unsigned int* errors_dev, void* stream);
/* The checking sampler (dd_tools_set_tuning(34, 2)): the lanes sampler with every thread's word of the mt19937 block mirrored in a register and
 * compared with the block in LDS at each member's start, before each regeneration, after each fill and at each member's end.  buf_dev:
 * dd_tools_sampler_dbg_words() zeroed uint32 words; afterwards word 0 != 0 = a check failed, word 1 = failed checks, words 2..12 = workgroup,
 * member, regenerations so far, check site, launch tag, read index, checks so far, LDS words dumped, HW_ID, XCC_ID, LDS_ALLOC; then five blocks
 * of 640 words from word 16: the registers' copy, the block as read, the block a few microseconds later, the state the launch loaded, (free);
 * then the workgroup's whole dynamic LDS. */
int dd_tools_sampler_dbg_attach(uint32_t* buf_dev);
size_t dd_tools_sampler_dbg_words(void);
unsigned int dd_tools_sampler_dbg_launches(void);
int dd_tools_hold_probe(int kind, int launches, int wgs, int hold, unsigned int* errors_dev, void* stream);
int dd_tools_lds_overlap_probe(int lds_a, int wgs_a, int hold_a, int lds_b, int wgs_b, int hold_b, int launches_b, unsigned int* errors_dev,
                               void* stream_a, void* stream_b);

/* Round 6 (prefill beside decode by CU partition, tools/cu_partition_lab.py): a HIP stream whose kernels run only on the CUs whose bits are
 * set in mask[0..words) (hipExtStreamCreateWithCUMask; 8 words = 256 CUs), and a probe that reports where workgroups launched on a stream
 * land: out_dev[2 i] = XCC_ID, out_dev[2 i + 1] = HW_ID of workgroup i, each holding its CU for `hold` rounds of s_sleep(64). */
int dd_tools_stream_create_cu_mask(const uint32_t* mask, int words, void** stream_out);
int dd_tools_stream_destroy(void* stream);
int dd_tools_cu_probe(uint32_t* out_dev, int wgs, int hold, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DROPDEC_TOOLS_H */
