/* libfdm_hip.so -- C ABI of the MI355X-native FDM diffusion-sampling hot path.
 *
 * Drop-in boundary.  The reference is pure Python and has no FFI; the "plugin
 * boundary" it offers is the Python class surface (FDM.forward, GaussianDiffusion.sample /
 * ddim_sample / p_sample, HubertModel.forward, VQAutoEncoder.quant / decode).  Every entry point
 * below names the reference interface (file:line under /root/reference) whose arithmetic it
 * replaces.  All functions take raw device pointers, sizes and a hipStream_t (passed as void*),
 * return 0 on success or a negative error code (message via fdm_last_error()) and never throw.
 * Nothing here takes or returns a torch type.
 *
 * Synchronisation contract, per layer:
 *   fdm_op_*, fdm_prog_run / _replay     enqueue on the given stream and return: they never synchronise
 *   fdm_prog_instantiate / _destroy      graph capture / release (host work; _destroy expects the stream drained)
 *   fdm_plan_commit, _reserve, _tune, _set("tile.*"), fdm_plan_set_weights after a commit, fdm_audio_prepare when it
 *     has to commit, grow the workspaces or tune               PLAN-TIME: allocate and drain the stream
 *   fdm_sample_graph, fdm_denoise_step   ONE stream drain per call (the timestep list / seed are host memory of the
 *     caller and are uploaded before the loop), plus graph instantiation the first time a program shape is used and a
 *     drain when the 8-entry program cache evicts; NOTHING synchronises inside the T-step loop
 *   fdm_hubert_forward, fdm_vq_*         drain the stream when they grow their workspaces (first call at a larger shape)
 *
 * Three layers:
 *   fdm_op_*    single-kernel operators (one launch on the given stream)
 *   fdm_prog_*  a recorded sequence of operators = one "step program", captured into a hipGraph and
 *               replayed T times with the diffusion timestep read from a device-side counter
 *   fdm_plan_* / fdm_audio_prepare / fdm_denoise_step / fdm_sample_graph
 *               the denoiser + scheduler of one model (weights by reference state-dict name, per-model and
 *               per-clip tables, workspaces, the step program and its hipGraph): what FDM.__init__ / FDM.forward
 *               and GaussianDiffusion.sample / ddim_sample do in the reference, behind plain pointers
 */
#ifndef FDM_HIP_H
#define FDM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define FDM_F32 0
#define FDM_BF16 1
/* Split operands (GEMM inputs only): x = hi + lo / S held as two planes of a 16-bit type, the lo plane `*_lo_off`
 * ELEMENTS after the hi plane, and the product evaluated as hi.hi + (hi.lo + lo.hi) / S in three passes of the 16-bit
 * MFMA with fp32 accumulation.  FDM_F16X3: fp16 planes, S = 2^11 (22 significant bits per operand: fp32-class results,
 * the mode that meets the 1e-4 contract on the 16-bit matrix cores; |x| is clamped to 65504); its QKV projection writes Q and
 * the packed K / V as plane pairs too and attention runs split (fdm_attn_args).  Every producer of a GEMM input writes the plane
 * pair.  (A bf16-plane split -- 16 bits per operand, ~5e-5 per denoiser call, outside the contract over chains -- was built and
 * measured in rounds 2-4 as kind 3 and is no longer part of the library: tools/sim_split_precision.py) */
#define FDM_F16X3 2
/* Single-plane fp16 operands (round 6): the `hi` plane of the split kind alone -- the bf16 kind's bytes and MFMA rate with 11
 * significand bits instead of 8 (|x| clamped to 65504 when an operand copy is written).  The denoiser's step program only
 * (fdm_op_gemm / fdm_op_attention / fdm_op_layernorm / fdm_op_cast / fdm_op_sched_step, fdm_plan_create); the audio encoders and the
 * VQ stages take FDM_F32 / FDM_BF16 / FDM_F16X3. */
#define FDM_F16 3

#define FDM_ACT_NONE 0
#define FDM_ACT_RELU 1       /* nn.TransformerDecoderLayer default activation, models/fdm_vocaset.py:45 */
#define FDM_ACT_MISH 2       /* nn.Mish, models/fdm_vocaset.py:22,31,38 */
#define FDM_ACT_GELU_ERF 3   /* transformers HuBERT 'gelu' */
#define FDM_ACT_GELU_TANH 4  /* models/utils/base_model_util.py:81-94 */
#define FDM_ACT_LEAKY02 5    /* nn.LeakyReLU(0.2), models/vq_vae_vocaset.py:206 */

#define FDM_OK 0
#define FDM_ERR_ARG (-1)
#define FDM_ERR_SHAPE (-2)
#define FDM_ERR_HIP (-3)
#define FDM_ERR_STATE (-4)

const char* fdm_last_error(void);
int fdm_version(void);
/* sizeof() of a public struct of this header ("fdm_gemm_args", "fdm_ln_args", ...) in the loaded build: lets a binding in
 * another language check its mirror of the struct before the first call (negative = unknown name) */
int fdm_abi_struct_size(const char* name);
/* 1 if a gfx950 device is visible to this process, else 0 (no device is touched otherwise) */
int fdm_device_ok(void);

/* ------------------------------------------------------------------------------------------
 * Scheduler (GaussianDiffusion.q_posterior + p_sample, ddim_sample update, CFG mix):
 * video_diffusion_pytorch/diffusion_BIWI_encoder_decoder.py:632-656, 693-708;
 * utiles/classifierfree.py:20-21.  All clips of the batch share the timestep (a3).
 * tables are fp32 [T_train] arrays; the current step k = *step (device int, incremented by the
 * kernel when advance != 0), t = tseq[k].  x0u != NULL enables the CFG mix
 * x0 = x0u + cfg_scale*(x0 - x0u) before the update.
 * DDPM: x' = c1[t]*x0 + c2[t]*x + sigma[t]*z, z = 0 when t == 0.  z comes from `noise`
 * (+ k*n elements) if non-NULL, else Philox4x32-10/Box-Muller keyed by (seed, clip0 + clip, k).
 * DDIM (eta = 0): eps = (sra[t]*x - x0)/srm1[t]; x' = x0*sqrt_an[k] + c_n[k]*eps.           */
typedef struct fdm_sched_args {
  const float* x0; const float* x0u; float cfg_scale;
  const float* x; float* x_out;
  long long n; long long n_per_clip;
  const int* tseq; int* step; int advance;
  const float* c1; const float* c2; const float* sigma;      /* DDPM tables, indexed by t */
  const float* sra; const float* srm1;                       /* DDIM tables, indexed by t */
  const float* sqrt_an; const float* c_n;                    /* DDIM tables, indexed by step k */
  const float* noise; long long noise_stride;               /* elements between steps (0 -> n) */
  void* x_out_t; int out_dtype;                              /* optional operand-dtype copy of x_out (next step's GEMM input) */
  unsigned int* arrive;                                      /* device word (zeroed once): lets the LAST block to read *step
                                                                advance it inside this kernel (no separate launch) */
  unsigned long long seed; int clip0;
  int mode;                                                  /* 0 DDPM, 1 DDIM, 2 CFG mix only (x_out = mix) */
  long long x_out_t_lo_off;                                  /* split out_dtype: elements between the hi and lo planes of x_out_t */
  const unsigned long long* seed_dev;                        /* optional device words {seed, clip0}: override `seed` / `clip0`, so a
                                                                captured graph serves every seed / shard (no re-capture per call) */
} fdm_sched_args;
int fdm_op_sched_step(const fdm_sched_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * C[M,N] = epilogue(A[M,K] * W[N,K]^T): every nn.Linear / Conv1d-as-GEMM on the path
 * (models/fdm_vocaset.py:20-24,36-39,45-51; transformers HubertAttention/FeedForward;
 * models/lib/base_models.py:71-87,138-174; models/vq_vae_vocaset.py:204,243).
 * A rows may overlap (lda < K) which expresses a strided Conv1d over a channels-last signal
 * without im2col.  v = acc + bias[n]; v = act(v); v += resid[m or m % mod][n]; stores fp32
 * and/or operand-dtype copies.  For a fused QKV projection the K columns [kp_col0, vp_col0) and the
 * V columns [vp_col0, N) can be written straight into the attention kernel's fragment-packed
 * buffers (see fdm_attn_args) instead of out_t/out_f32.  K must be a multiple of 32 (fp32) /
 * 64 (16-bit kinds); A, W 16-byte aligned with lda, ldw multiples of 4 (fp32) / 8 (16-bit kinds), 0 <= lda, ldw < 2^31 (they reach
 * the kernel as 32-bit preloaded arguments; FDM_ERR_SHAPE otherwise).   */
typedef struct fdm_gemm_args {
  const void* A; long long lda; long long a_batch_stride;
  const void* W; long long ldw; long long w_batch_stride;
  int M, N, K, batch;
  int dtype;                      /* FDM_F32 | FDM_BF16: type of A, W, out_t, out_kp, out_vp;
                                     FDM_F16X3: A, W, out_t, out_kp, out_vp are fp16 plane pairs */
  const float* bias; long long bias_batch_stride;
  int act;
  const float* resid; long long ldr; int resid_row_mod;
  float* out_f32; long long ldo_f32;
  void* out_t; long long ldo_t;
  long long out_batch_stride;     /* elements, applied to out_f32, out_t and resid (column offset) */
  /* packed K / V outputs: rows are (clip b = m / kv_L, key l = m % kv_L), columns h*kv_hd + e inside each range */
  void* out_kp; int kp_col0;
  void* out_vp; int vp_col0;
  int kv_L; int kv_Lpad; int kv_hd;
  /* --- LayerNorm folded into the GEMMs around it (bf16 step program; removes the norm3 launch) ---
   * producer: stat_out != NULL -> per-row partial (sum v, sum v^2) of the fp32 outputs of each 64-column
   *   group are written to stat_out[(n/64) * 2M + 2m + {0,1}] (plain stores; summed per 16-column fragment, fragments in
   *   column order: deterministic AND independent of the output tile, so `tile` never changes results).
   * consumer: ln_stat_in != NULL -> mu_m, rstd_m = f(sum over ln_nparts partials, ln_dim, ln_eps).
   *   ln_colsum != NULL: A holds the RAW (un-normalised) rows and W = W o gamma, so
   *     LN(x) W^T + b  ==  rstd_m (acc - mu_m colsum_n) + bias_n   with bias_n := beta.W_n + b_n;
   *   rln_gamma != NULL: `resid` holds the raw rows and the residual added is LN(resid) computed on the fly. */
  float* stat_out;
  const float* ln_stat_in; int ln_nparts; int ln_dim; float ln_eps;
  const float* ln_colsum;
  const float* rln_gamma; const float* rln_beta;
  /* optional device int incremented once (by one thread) when the kernel starts: the first GEMM of a diffusion
   * step advances the device-side step counter this way (no extra launch, no atomics on a hot word) */
  int* incr_counter;
  const int* incr_table;          /* optional: incr_counter[1] = incr_table[new counter value] (t = tseq[step] for the step) */
  /* output tile per workgroup: 0 = library heuristic, else FDM_TILE_*.  Results do not depend on it (every tile
   * accumulates k in the same order): callers time the candidates once per shape at plan build and pass the winner. */
  int tile;
  /* sched_fuse != 0: the epilogue applies fdm_op_sched_step's DDPM / DDIM update (sched.mode 0 / 1, no CFG mix) to the
   * tile it just computed, v = x0_hat: `resid` is read as the current latent x_t (NOT added), and x_{t-1} goes to
   * out_f32 (may alias resid) and, if given, out_t (the next step's operand copy).  The latent-decoder GEMM of a
   * non-CFG sampler uses this: one launch less per diffusion step, bit-identical to the separate kernel.  Needs
   * N % 64 == 0, ldo_f32 == ldr == N, 16-byte aligned pointers; sched.x0 / x0u / x / x_out / x_out_t / arrive unused. */
  int sched_fuse;
  fdm_sched_args sched;
  /* split operand kind (dtype FDM_F16X3): elements between the hi and lo planes of A, W and out_t */
  long long a_lo_off, w_lo_off, out_t_lo_off;
  long long kv_lo_off;            /* FDM_F16X3 with out_kp / out_vp: elements between the hi and lo planes of the packed buffers */
  /* --- split K (round 5): ksplit = S > 1 runs S workgroups per output tile (blockIdx.z = slice s); slice s accumulates the
   * k-tiles [s nk / S, (s + 1) nk / S) in the usual order and stores its fp32 partial tile to out_f32 + s * ksplit_stride
   * (elements); slice 0 carries bias and residual, the others neither.  The consumer sums the S planes in plane order
   * (fdm_ln_args.x_planes): no atomics, no extra launch, deterministic.  A workgroup's dependent k chain is 1/S as long and S
   * chains share a CU.  Results depend on S (the k order changes), never on `tile`.  Needs batch <= 1, act NONE, out_f32 as the
   * only output, dense vectorisable rows (N % 64 == 0), (K / 64 [16-bit kinds] or K / 32 [fp32]) % S == 0, S <= 4 (8 measured: no further gain); tiles: the
   * 64-column tiles FDM_TILE_64x64, FDM_TILE_64x64_S2 and FDM_TILE_32x64_S3. */
  int ksplit; long long ksplit_stride;
  /* --- second batch level (round 5): batch2 = C >= 1 runs C x batch problems in one launch, z = (c, g): A advances by
   * a_batch_stride per g and by a_batch_stride2 per c, outputs and resid by out_batch_stride (a column offset) per g and by
   * out_batch_stride2 (elements: a row offset) per c; W and bias depend on g alone.  This is a grouped Conv1d over C clips
   * (HuBERT's positional conv: g = channel group, c = clip).  When batch % 8 == 0 the workgroups are dealt so that XCD x serves
   * the groups [x batch / 8, (x + 1) batch / 8) only, group-major: each L2 streams its 1 / 8 of W once for all clips and row
   * tiles.  Tiles: FDM_TILE_64x64, FDM_TILE_64x64_S2 and FDM_TILE_128x64; no ksplit, no LayerNorm folds, no packed K / V, no fused scheduler. */
  int batch2; long long a_batch_stride2, out_batch_stride2;
} fdm_gemm_args;
#define FDM_TILE_AUTO 0
/* Eight tiles (round 5; eleven before).  Every tile accumulates k in the same order: the choice changes speed, never results. */
#define FDM_TILE_64x64 1       /* 8 waves, 4-stage ring (64 KB; split kinds 128 KB) */
#define FDM_TILE_128x64 2      /* 4-stage ring while the launch is one round (<= 256 workgroups), else 3-stage (72 KB: two workgroups per CU);
                                  split kinds: 3-stage (144 KB) */
#define FDM_TILE_128x128 3
#define FDM_TILE_64x64_S2 8    /* 64x64 with a 2-stage ring (four workgroups per CU; split kinds two) */
#define FDM_TILE_32x64_S3 9    /* 32x64 on 4 waves, 3-stage ring: twice the workgroups of 64x64 for few-hundred-row GEMMs */
#define FDM_TILE_256x128_PP 10 /* 256x128, two wave groups half a period apart (one computes while the other loads): large M; split kinds: 128x128 */
#define FDM_TILE_80x128 11     /* 80x128: ten row tiles for 800 rows -> 240 workgroups at N = 3072 (the QKV projection of four 200-frame clips) */
#define FDM_TILE_64x128 12     /* 64x128: 13 row tiles for 800 rows -> 208 workgroups at N = 2048 in one round (FFN1 in the split modes) */
/* retired ids (accepted, resolve to a live tile; no heuristic rule returns them and the tuner does not time them):
 *   4  96x128 (round 4: never picked on 66 shapes)                         -> 128x128
 *   5  256x128 on the lockstep loop (round 5: 29.9 us against 29.4 for the ping-pong loop on its one heuristic site) -> 256x128_PP
 *   6  64x64 on a 3-stage ring (round 5: 3 tuner picks on 66 shapes, no heuristic rule) -> 64x64
 *   7  128x64 on a 3-stage ring (round 5: the ring depth follows the grid, see 128x64)  -> 128x64 */
#define FDM_TILE_96x128 4
#define FDM_TILE_256x128 5
#define FDM_TILE_64x64_S3 6
#define FDM_TILE_128x64_S3 7
#define FDM_TILE_MAX 12
/* or-ed into `tile`: run the general (edge-handling) kernel even where a specialised one would do -- the two must agree bit for
 * bit (tests/test_ops_gpu.py); not a tuning knob.  Honoured by every tile including the ping-pong one; the scheduler-fused latent
 * decoder (sched_fuse) has lean kernels only and ignores it. */
#define FDM_TILE_GENERAL 0x100
/* or-ed into `tile`: the lockstep k loop (every wave issues its own tile loads) instead of the loader-wave form the 16-bit kinds run
 * by default (round 5) -- the two are bit-identical (tests/test_ops_gpu.py); A/B measurements and tests, not a tuning knob */
#define FDM_TILE_LOCKSTEP 0x200
#define FDM_TILE_ID_MASK 0xff
int fdm_op_gemm(const fdm_gemm_args* a, void* stream);
/* The FDM_TILE_* value a launch of *a with tile = 0 resolves to (the library heuristic on M, N, K, batch and the operand kind;
 * no device work, a->tile is ignored).  The plan-time tuner uses it to leave the heuristic's own tile out of its candidates. */
int fdm_gemm_heuristic_tile(const fdm_gemm_args* a);

/* ------------------------------------------------------------------------------------------
 * Fused softmax(Q K^T * scale + bias) V for one [B, H, L, hd] problem.
 * nn.MultiheadAttention self-attention with the causal periodic-ALiBi mask generated in-kernel
 * (models/fdm_vocaset.py:85,95-116: mask[h,i,j] = -slope_h*floor((i-j)/period), -inf for j>i);
 * non-causal for HuBERT (hd 64, scale 1/8) and the VQ decoder (hd 128, scale hidden^-0.5,
 * models/lib/base_models.py:144).  Q: row (b*L + l), column h*hd + e, row stride ldq.
 * O: [B*L, ldo] operand dtype.
 * Kp, Vp: "fragment-packed" keys / values, one block of Lpad*hd elements per (b, h) (Lpad a
 * multiple of 32, pad keys must hold finite values, e.g. zeros), laid out so that every MFMA
 * operand fragment of a key tile is ONE contiguous 1 KB run (64 lanes x 16 B):
 *   EPC = 16 / sizeof(T) elements per chunk, key tile KT = 4*EPC keys (bf16 32, fp32 16)
 *   Kp[ ((((kt*NSUB + s)*NKS + ks)*4 + g)*16 + r)*EPC + e%EPC ]   NSUB = KT/16, NKS = hd/(4*EPC)
 *       key l = kt*KT + w;  bf16: s = (w>>2)&1, r = 4*(w>>3) + (w&3);  fp32: s = 0, r = w
 *       column e: chunk e/EPC = 4*ks + g
 *   Vp[ (((kt*(hd/16) + e/16)*4 + g)*16 + e%16)*EPC + w%EPC ]        g = w / EPC
 * Written by fdm_op_gemm (out_kp / out_vp) or by fdm_op_pack_kv from row-major K, V.          */
typedef struct fdm_attn_args {
  const void* Q; long long ldq;     /* ldq, q_lo_off, kv_lo_off < 2^31: 32-bit preloaded kernel arguments (FDM_ERR_SHAPE otherwise) */
  const void* Kp;
  const void* Vp; int Lpad;
  void* O; long long ldo;
  int B, H, L, hd;
  int dtype;
  float scale;
  int causal;
  const float* slopes;   /* [H] device floats or NULL */
  int period;
  /* o_split = FDM_F16X3 (with dtype FDM_F32): O is written as a split plane pair (the next GEMM's input),
   * lo plane o_lo_off elements after the hi plane; 0 = O has the dtype of Q */
  int o_split; long long o_lo_off;
  /* dtype FDM_F16X3: Q, Kp, Vp and O are fp16 plane pairs (the 16-bit packed layout, per plane); both products run as three
   * 16-bit MFMA passes with the probabilities split in registers: fp32-class results.  q_lo_off / kv_lo_off / o_lo_off =
   * elements between the hi and lo planes of Q / Kp and Vp / O. */
  long long q_lo_off, kv_lo_off;
} fdm_attn_args;
int fdm_op_attention(const fdm_attn_args* a, void* stream);
/* row-major K, V (row b*L + l, column h*hd + e, row strides ldk / ldv) -> the packed layouts above */
int fdm_op_pack_kv(const void* K, long long ldk, const void* V, long long ldv, void* Kp, void* Vp,
                   int B, int H, int L, int Lpad, int hd, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * y = act(LayerNorm(x + add_mat + add_tab[idx]) * gamma + beta), one wavefront per row
 * (nn.LayerNorm eps 1e-5: decoder norm1-3 models/fdm_vocaset.py:45; HuBERT; VQ Norm
 * models/lib/base_models.py:37-52).  add_tab row index = tab_index[*tab_step] if tab_step else
 * tab_index[0] (device ints) -- this is how the folded cross-attention time term enters
 * (SURVEY.md a11x).  d in {256, 512, 768, 1024}.                                                  */
typedef struct fdm_ln_args {
  const float* x; int M, d;
  const float* add_mat;               /* [M, d] or NULL */
  const float* add_tab;               /* [rows, d] or NULL */
  const int* tab_index; const int* tab_step;
  const float* gamma; const float* beta; float eps;
  int act;
  float* y_f32; void* y_t; int dtype;
  /* optional fused second LayerNorm (post-norm decoder: norm1 then norm2 back to back,
   * models/fdm_vocaset.py:45): y = LN2(LN1(x) + add_mat + add_tab[idx]); the addends then belong to
   * stage 2 and stage 1 sees x alone. */
  const float* gamma2; const float* beta2;
  long long y_t_lo_off;               /* split dtype: elements between the hi and lo planes of y_t */
  /* add_mat shared by the S conditions of a clip (fdm_audio_prepare_conds): add_mat_group > 0 -> row m reads add_mat row
   * ((m % add_mat_wrap) / add_mat_group) * add_mat_L + m % add_mat_L, i.e. rows are [wrap block][clip][condition][frame]
   * (add_mat_group = S * L, add_mat_L = L, add_mat_wrap = rows per cond/uncond half, 0 = no wrap) over an add_mat of
   * [clips * L, d]; 0 = row m reads add_mat row m */
  int add_mat_L, add_mat_group, add_mat_wrap;
  /* x given as x_planes (2..4) partial planes, x_plane_stride elements apart (a split-K GEMM's outputs, fdm_gemm_args.ksplit):
   * the row is ((x[0] + x[1]) + x[2]) + ..., summed in plane order before anything else; 0 / 1 = one plane */
  int x_planes; long long x_plane_stride;
} fdm_ln_args;
int fdm_op_layernorm(const fdm_ln_args* a, void* stream);


/* ------------------------------------------------------------------------------------------
 * Small elementwise / layout operators.                                                      */
/* dst_t[i] = (dtype) src_f32[i]; split dtypes: hi plane at dst, lo plane at dst + n elements */
int fdm_op_cast(const float* src, void* dst, long long n, int dtype, void* stream);
/* out[r, :] = act(in[r, :] + vec[:]) -- e.g. tau table = Mish(W_t^T + b_t), models/fdm_vocaset.py:71-72 */
int fdm_op_bias_act(const float* in, const float* vec, float* out, long long rows, int d, int act, void* stream);
/* out[m, :] = a[(m / a_div) % a_mod, :] (+ b[(m / b_div) % b_mod, :]) (+ c[...]) -- conditioning addend
 * table: PE[l] + style[b] (+ emotion[b]), models/fdm_vocaset.py:75-84 */
int fdm_op_add_rows(const float* a, int a_div, int a_mod, const float* b, int b_div, int b_mod,
                    const float* c, int c_div, int c_mod, float* out, long long M, int d, void* stream);
/* out[b, :] = act(W[d, K] x[b, :] + bias) for conditioning one-hots (K <= 64): style_embedd /
 * emotion_embedd, models/fdm_vocaset.py:34,75; models/fdm_vqvae_mead.py:34-36,85 */
int fdm_op_small_linear(const float* x, const float* W, const float* bias, float* out, int B, int K, int d,
                        int act, void* stream);
/* out[b, k, :] = in[b, clamp(k - pad, 0, L-1), :] for k in [0, L + 2*pad): replicate padding, channels-last */
int fdm_op_pad_rows(const void* in, void* out, int B, int L, int d, int pad, int dtype, int zero, void* stream);
/* HuBERT / wav2vec2 conv layer 0: wav [B, n] -> out [B, T0, 512], Conv1d(1, 512, k=10, s=5) (+ bias if non-NULL) */
int fdm_op_conv0(const float* wav, const float* w, const float* bias, float* out, int B, int n, int T0, void* stream);
/* the same layer with the LayerNorm(512, eps) + GELU(erf) that follows it in HuBERT-large (feat_extract_norm = 'layer') applied
 * before the store: out [B, T0, 512] in fp32 or bf16 (dtype), nothing else written */
int fdm_op_conv0_ln_gelu(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, void* out,
                         long long out_lo_off, int B, int n, int T0, float eps, int dtype, void* stream);     /* dtype FDM_F16X3: a plane pair, lo plane out_lo_off elements on */
/* per-(clip, channel) InstanceNorm1d over L after LeakyReLU(0.2): models/vq_vae_vocaset.py:204-209 */
int fdm_op_leaky_instnorm(const float* x, float* y_f32, void* y_t, int B, int L, int d, float eps, int dtype, void* stream);
/* GroupNorm(num_groups = C, affine) over time + activation, channels-last x [B, T, C]: first conv layer of
 * wav2vec2-base (transformers Wav2Vec2GroupNormConvLayer; BIWI audio encoder, models/wav2vec.py:69-143) */
/* scratch (optional, 8-byte aligned, >= B * min(64, ceil(T / 1024)) * C * 16 bytes): with it, clips of T >= 4096 frames are
 * normalised over time chunks in two launches of hundreds of workgroups (fp64 chunk statistics folded in chunk order); without
 * it one launch of C / 64 workgroups per clip does the three passes (fine for short clips; the operator never allocates) */
int fdm_op_time_groupnorm(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_t, long long y_t_lo_off, int B, int T, int C,
                          float eps, int act, int dtype, void* scratch, long long scratch_bytes, void* stream);
/* out[0] = mean(|a - b|^p), p = 2 (l1 = 0) or 1: the forward value of p_losses' F.mse_loss / F.l1_loss
 * (diffusion_BIWI_encoder_decoder.py:744-749); partial: >= 1024 floats of scratch; deterministic order */
int fdm_op_mean_diff(const float* a, const float* b, float* partial, float* out, long long n, int l1, void* stream);
/* Evaluation metrics over vertex sequences gt, pred [F, V, 3] fp32 (computer_metrix.py:84-136, metric/metric.py:115-138).
 * region: R int32 vertex indices (NULL with R == V: all vertices).  Per frame: frame_max[f] = max_r d2 (bit-identical to
 * numpy's float32 value), frame_sum[2f] = sum_r d2, frame_sum[2f+1] = sum_r sqrt(d2) with d2 = |gt - pred|^2;
 * out[0] = mean_f frame_max ("Lip/Face Vertex Error"), out[1] = mean d2 ("Emotion Mean Error"), out[2] = mean |d|
 * ("Mean Vertex Error"; compute_diversity's pairwise distance).  Deterministic (fixed reduction order).                  */
int fdm_op_vertex_err(const float* gt, const float* pred, const int* region, int R, int F, int V,
                      float* frame_max, double* frame_sum, double* out, void* stream);
/* out[0] = mean_r std_f(|verts[f, region[r]] - tmpl[region[r]]|^2): the per-sequence motion statistic whose gt - pred
 * difference is FDD (computer_metrix.py:95-107).  partial: >= 2 * min(F, 64) * R doubles of scratch.                     */
int fdm_op_motion_std(const float* verts, const float* tmpl, const int* region, int R, int F, int V,
                      double* partial, double* out, void* stream);
/* F.interpolate(mode='linear', align_corners=True) over time, x [B, Tin, C] -> y [B, Tout, C] fp32
 * (linear_interpolation, models/hubert.py:62-69: optional 50 -> 30 fps resampling of the conv features)                  */
int fdm_op_linear_interp(const float* x, float* y, int B, int Tin, int Tout, int C, void* stream);
/* AdaIN (utiles/adaIN.py:4-22): content, style [N, C, Lc], [N, C, Ls] -> out [N, C, Lc] */
int fdm_op_adain(const float* content, const float* style, float* out, int NC, int Lc, int Ls, float eps, void* stream);
/* regroup [B, T, d] -> [groups, B, T + 2*pad, d/groups] zero padded (HuBERT positional conv input) */
int fdm_op_group_pad(const void* in, void* out, int B, int T, int d, int groups, int pad, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * VectorQuantizer.forward (models/lib/quantizer.py:35-64, models/vq_vae_emotion.py:221-252):
 * d_k = (sum z^2 + sum e_k^2) - 2 z.e_k, first-min argmin, z_q = z + (e - z), output
 * permuted to [B, c, R] (R = L*G rows per clip).  book[b] selects the 256-code slice.        */
int fdm_op_vq_quant(const float* z, const float* codebook, const int* book, int B, int R, int c, int K,
                    float* zq_bcl, long long* idx, void* stream);
/* The rest of VectorQuantizer.forward's return tuple (models/lib/quantizer.py:46-61; EVQ models/vq_vae_emotion.py:232-249) for the
 * indices idx [B*R] fdm_op_vq_quant chose: min_encodings [B*R, K] one-hot fp32 (optional), out[0] = loss =
 * beta * mean((e - z)^2) + mean((e - z)^2), out[1] = perplexity = exp(-sum_k p_k log(p_k + 1e-10)), p = column means of
 * min_encodings.  partial: >= 1024 doubles, hist: K ints of scratch.  Deterministic (fixed reduction order, integer histogram). */
int fdm_op_vq_stats(const float* z, const float* codebook, const int* book, const long long* idx, int B, int R, int c, int K, float beta,
                    float* min_encodings, double* partial, int* hist, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Step programs: record fdm_op_* calls, run them eagerly or as a hipGraph replayed n times.  */
typedef struct fdm_prog fdm_prog;
int fdm_prog_create(fdm_prog** out);
int fdm_prog_destroy(fdm_prog* p);
int fdm_prog_begin(fdm_prog* p);            /* subsequent fdm_op_* calls on this thread are recorded, not launched */
int fdm_prog_end(fdm_prog* p);
int fdm_prog_run(fdm_prog* p, void* stream);                 /* eager: launch every recorded op */
int fdm_prog_instantiate(fdm_prog* p, void* stream);         /* capture into a hipGraph */
int fdm_prog_replay(fdm_prog* p, int n, void* stream);       /* launch the graph n times */
int fdm_prog_num_ops(fdm_prog* p);

/* ------------------------------------------------------------------------------------------
 * Plan layer.  A plan owns device copies of the model's weights, the tables derived from them,
 * its workspaces and the recorded step program; the caller owns every tensor it passes in.  Device memory is allocated only
 * by fdm_plan_create / _reserve / _commit (and by fdm_audio_prepare when it has to grow the workspaces or commit first);
 * fdm_sample_graph / fdm_denoise_step drain the stream once per call while uploading the timestep list (host memory of the
 * caller) and instantiate a graph the first time a program shape is used; nothing synchronises inside the T-step loop.
 * A plan is not thread-safe; one plan per device/stream.
 *
 * Model geometry = the reference constructors' numbers (models/fdm_vocaset.py:9-51, models/fdm_vqvae_mead.py:9-53,
 * models/fdm.py:10-48, models/utils/config.py): fdm_model_preset fills it for "vocaset", "mead", "biwi" (+ "_tiny" test twins). */
typedef struct fdm_model_desc {
  int d, n_head, n_layers, ffn;   /* feature_dim, heads, decoder layers, dim_feedforward (= 2 d) */
  int G, c;                       /* latent vectors per frame, their width (G * c == d) */
  int n_style, n_emo;             /* one-hot widths of style_embedd / emotion_embedd (0 = no emotion input) */
  int audio_in, pair;             /* input width of audio_extract.0; audio-encoder frames folded per latent frame */
  int pe_periodic, period;        /* PeriodicPositionalEncoding (period) or plain sinusoidal PE; ALiBi period */
  int latent_mish, style_mish;    /* Mish after latent_encoder / style_embedd */
  int max_len;                    /* init_biased_mask(max_seq_len = 600), models/fdm_vocaset.py:44 */
} fdm_model_desc;
int fdm_model_preset(const char* name, fdm_model_desc* out);

typedef struct fdm_plan fdm_plan;
/* Workspaces for up to B clips x L latent frames (x2 rows when cfg != 0: cond + uncond rows in one set of launches).
 * dtype: FDM_F32 | FDM_BF16 | FDM_F16X3 = the arithmetic mode of the step program. */
int fdm_plan_create(const fdm_model_desc* desc, int B, int L, int cfg, int dtype, fdm_plan** out);
int fdm_plan_reserve(fdm_plan* p, int B, int L, int cfg);           /* grow the workspaces (allocates; drops recorded programs) */
int fdm_plan_destroy(fdm_plan* p);
/* One fp32 tensor of the reference state dict (FDM.state_dict() names, e.g. "transformer_decoder.layers.0.linear1.weight",
 * "latent_decoder.bias", optional buffer "PE.pe"), n elements at ptr (host or device memory); copied into the plan.
 * Optional schedule overrides "sched.c1", "sched.c2", "sched.sigma", "sched.sra", "sched.srm1" ([1000] fp32 each: the
 * caller's own q_posterior / predict_noise tables; default = fdm_schedule_host's). */
int fdm_plan_set_weights(fdm_plan* p, const char* name, const float* ptr, long long n, void* stream);
/* Per-model tables: operand-kind weight copies, tau[t] = Mish(W_t[:, t] + b_t), folded cross-attention time tables
 * TT_l = Wo_l Wv_l tau, LayerNorm folds of the bf16 program.  Called implicitly by fdm_audio_prepare. */
int fdm_plan_commit(fdm_plan* p, void* stream);
/* Once per batch of clips (the hoisted, step-invariant part of FDM.forward, models/fdm_vocaset.py:59-84):
 * hub [B, N, fw] audio-encoder features (fw * pair == audio_in), style [B, n_style], emo [B, n_emo] or NULL, device fp32.
 * L <= N / pair latent frames.  Builds AF = audio_extract(hub), the per-layer tables C1_l = Wo_l (Wv_l AF + bv_l) + bo_l
 * and E0 = PE + style (+ emotion; the uncond rows of a CFG plan use emotion_embedd's bias only).                        */
int fdm_audio_prepare(fdm_plan* p, const float* hub, int B, int N, int fw, const float* style, const float* emo,
                      int L, int cfg, void* stream);
/* The same for S conditions per clip: the reference's sampler loops the style one-hots of a clip through ddim_sample one
 * B = 1 call at a time with the SAME audio (samples/sample_diffusion_vocaset.py:71-83; for 3D-MEAD any set of
 * (emotion, identity) pairs of one utterance).  Here the S conditions of each of the B clips run as B*S rows blocks of ONE step
 * program: row block (b, s) = "virtual clip" b*S + s.  style [B*S, n_style], emo [B*S, n_emo] or NULL (virtual-clip order);
 * AF and the C1_l tables are computed once per CLIP (B*L rows) and shared by its S conditions -- only E0 is per condition.
 * Afterwards the plan's batch is B*S clips: x_T / out / noise of fdm_sample_graph and fdm_denoise_step are
 * [B*S, L*G, c], Philox noise is keyed by clip0 + b*S + s, and every row block is bit-identical to the B = 1, S = 1 call
 * with that clip's audio and that condition.  S = 1 is fdm_audio_prepare. */
int fdm_audio_prepare_conds(fdm_plan* p, const float* hub, int B, int N, int fw, int S, const float* style, const float* emo,
                            int L, int cfg, void* stream);
/* One FDM.forward (models/fdm_vocaset.py:54-91): x_t [B, L*G, c] -> x0_hat [B, L*G, c], CFG-mixed
 * (x0u + cfg_scale (x0 - x0u), utiles/classifierfree.py:20-21) when prepared with cfg.  x0_uncond (optional) receives
 * the unconditional rows.  Eager launches of the recorded step program. */
int fdm_denoise_step(fdm_plan* p, const float* x_t, int t, float cfg_scale, float* x0_hat, float* x0_uncond, void* stream);
/* GaussianDiffusion.p_sample_loop / ddim_sample (diffusion_BIWI_encoder_decoder.py:649-710, diffusion_mead_encoder_decoder.py:
 * 649-667): the step program (denoiser + fused scheduler update) replayed n_steps times as a hipGraph, the timestep read
 * from a device-side counter; graph_steps (default 10) diffusion steps are captured per graph launch. */
typedef struct fdm_sample_args {
  int kind;                       /* 0 = DDPM over t_list, 1 = DDIM (eta = 0) with `ddim_steps` (the dead last pair is skipped) */
  const float* x_T; float* out;   /* [B, L*G, c] device fp32 (may alias) */
  const int* t_list; int n_steps; /* DDPM: host array of timesteps, descending */
  int ddim_steps;
  const float* noise;             /* DDPM: [n_steps, B, L*G, c] device fp32 injected z, or NULL: Philox(seed, clip0 + b, step) */
  unsigned long long seed; int clip0;
  float cfg_scale;
  int eager;                      /* != 0: launch the recorded ops directly instead of replaying the graph (bit-identical) */
  float* record;                  /* optional [n_steps, B, L*G, c]: the latent after every step */
  int graph_steps;                /* diffusion steps per graph launch; 0 = default */
} fdm_sample_args;
int fdm_sample_graph(fdm_plan* p, const fdm_sample_args* a, void* stream);
/* Plan-time tuning of the GEMM output tiles at the prepared shape (times candidates per call site; changes speed only, every
 * tile accumulates k in the same order).  This call is the ONLY place the library tunes by itself: request paths
 * (fdm_audio_prepare*, fdm_sample_graph) never do -- fdm_plan_get(p, "needs_tune") turns 1 once the prepared shape has served
 * >= 2000 diffusion steps on heuristic tiles, so a serving caller can schedule this call off the request path.  Opt-in
 * (fdm_plan_set(p, "tune_lazy", 1)): tune inside fdm_audio_prepare* / fdm_sample_graph once that is the case; a tuner failure
 * there keeps the heuristic tiles and is counted ("tune_failed"), it never fails the request.
 * The library reads five environment variables, all about tiles: FDM_TUNE=0 disables the tuner (heuristic tiles, or the pinned
 * set); FDM_TUNE_VERBOSE=1 prints its choices; FDM_TILE_OVERRIDE="qkv=3,ffn1=2" pins call sites (applied at fdm_audio_prepare*,
 * with or without the tuner); FDM_TILE_CACHE=<file> (opt-in) keeps tuned sets across processes: a tuning run appends
 * "<version|mode|geometry|shape>\t<site>=<tile>,..." (temporary file + rename), and fdm_audio_prepare* takes a stored set for its
 * shape without any timing launch; FDM_GEMM_TILE=<FDM_TILE_*> forces one tile for every fdm_op_gemm with tile = 0 (A/B sweeps);
 * FDM_GEMM_LOCKSTEP=1 = FDM_TILE_LOCKSTEP for every fdm_op_gemm of the process (A/B of the once-per-clip stages; the step has fdm_plan_set "lockstep"). */
int fdm_plan_tune(fdm_plan* p, void* stream);
/* Introspection / experiments: integer properties by name -- "launches_per_step", "graph_launches" (host graph launches of
 * the last fdm_sample_graph), "rows", "tuned", "needs_tune", "tune_failed", "fuse_ln3", "tile.<call site>" (qkv, out, ffn1, ffn2,
 * enc, dec, ...). */
int fdm_plan_get(fdm_plan* p, const char* key, long long* out);
/* "tile.<call site>" (drops recorded programs), "tune" (0 = off), "tune_lazy" (1 = in-call tuning allowed), "untune" (forget every
 * tuned set), "fuse_ln3" (1 = fold norm3 into the GEMMs around it: 8 launches fewer per step, no longer faster; takes effect at
 * the next commit) */
int fdm_plan_set(fdm_plan* p, const char* key, long long value);

/* ------------------------------------------------------------------------------------------
 * Audio encoder (once per clip: it does not depend on (t, x_t), so the reference's per-step re-run, models/fdm_vocaset.py:59,
 * is hoisted).  kind 0 = HuBERT-large (models/hubert.py:75-146: 7 x {Conv1d, LayerNorm(512), GELU}, even crop, projection,
 * weight-normed grouped positional conv, 24 pre-LN layers, final LayerNorm); kind 1 = wav2vec2-base (models/wav2vec.py:
 * 69-143: GroupNorm on conv 0 only, no conv bias, 12 post-LN layers) -- the BIWI denoiser's encoder.  n_layers 0 = the
 * kind's depth.  Weights by transformers state-dict name ("feature_extractor.conv_layers.0.conv.weight", ...,
 * "encoder.pos_conv_embed.conv.parametrizations.weight.original0|1" or torch-2.0's "...conv.weight_g|weight_v").
 * forward: wav [B, n] processor-normalised fp32 -> out [B, N, D] fp32, *n_frames = N = fdm_hubert_frames(n) (the conv
 * stack's length, even-cropped, :95-96).  frame_num > 0 keeps at most 2 * frame_num frames (:97-98); interp_in_fps /
 * interp_out_fps > 0 resample the conv features (linear_interpolation, :62-69) to frame_num or int(T / in * out) frames
 * instead of the even crop.  The caller sizes `out` for fdm_hubert_frames(n) rows per clip (or the interpolated count).
 * dtype: FDM_F32, FDM_BF16, or FDM_F16X3 = the transformer layers (84 % of the FLOPs) on split-fp16 operands behind an fp32
 * conv front / projection / positional conv: inside the 1e-4 contract at 0.57x the fp32 encoder's time. */
typedef struct fdm_audio_encoder fdm_audio_encoder;
int fdm_hubert_create(int kind, int n_layers, int dtype, fdm_audio_encoder** out);
int fdm_hubert_set_weights(fdm_audio_encoder* e, const char* name, const float* ptr, long long n, void* stream);
int fdm_hubert_forward(fdm_audio_encoder* e, const float* wav, int B, int n_samples, int frame_num, int interp_in_fps, int interp_out_fps,
                       float* out, int* n_frames, void* stream);
int fdm_hubert_frames(int n_samples);
int fdm_hubert_destroy(fdm_audio_encoder* e);

/* ------------------------------------------------------------------------------------------
 * (E)VQ-VAE: quantise, decode, encode (models/vq_vae_vocaset.py:23-43, models/vq_vae_emotion.py:9-41, models/vq_vae.py).
 * Geometry from the reference's *_vq_vae_args (models/utils/config.py): G = face_quan_num, c = zquant_dim, K = 256 codes
 * per book, n_books = n_embed / 256 (emotion-sliced codebook), V3 = in_dim; pre = decoder_linear_embedding_pre /
 * encoder_linear_embedding_post present (3D-MEAD, BIWI).  Weights by reference state-dict name.
 * quant:  z [B, R, c] (+ emotion one-hot [B, n_books]) -> z_q [B, c, R] (the reference's permuted output), idx [B*R] int64;
 *         quant_stats: emb_loss, perplexity, min_encodings of that call
 * decode: z_q [B, c, L*G] -> vertex offsets [B, L, V3] (the caller adds the template; every clip gets pe[0], a20)
 * encode: x [B, L, V3] (+ emotion one-hot [B, 7]) -> latent [B, L*G, c]
 * dtype: FDM_F32, FDM_BF16, or FDM_F16X3 = the two 6-layer transformers on split-fp16 operands, convolutions / embeddings /
 *        vertex map / quantiser in fp32 (indices and z_q identical to the fp32 object's; decode 1.4-4.6e-5 from the reference) */
typedef struct fdm_vq_desc { int G, c, K, n_books, V3, pre; } fdm_vq_desc;
typedef struct fdm_vq fdm_vq;
int fdm_vq_create(const fdm_vq_desc* desc, int dtype, fdm_vq** out);
int fdm_vq_set_weights(fdm_vq* v, const char* name, const float* ptr, long long n, void* stream);
int fdm_vq_quant(fdm_vq* v, const float* z, const float* emo_one_hot, int B, int R, float* zq_bcl, long long* idx, void* stream);
/* emb_loss, perplexity and min_encodings of the same call (the reference returns them from quant(): models/vq_vae_vocaset.py:31-33,
 * beta = 0.25 :16-18): out2 = {loss, perplexity} device floats, min_encodings [B*R, K] device fp32 or NULL. */
int fdm_vq_quant_stats(fdm_vq* v, const float* z, const float* emo_one_hot, const long long* idx, int B, int R, float beta,
                       float* min_encodings, float* out2, void* stream);
int fdm_vq_decode(fdm_vq* v, const float* zq_bcl, int B, int R, float* out, void* stream);
int fdm_vq_encode(fdm_vq* v, const float* x, const float* emo_one_hot, int B, int L, float* latent, void* stream);
int fdm_vq_destroy(fdm_vq* v);

/* Host-side tables (no device needed): the 12 GaussianDiffusion buffers in registration order, T floats each
 * (diffusion_BIWI_encoder_decoder.py:565-603: cosine schedule in fp64, fp32 cast); DDIM pairs and per-pair coefficients
 * (:684-708, eta = 0; returns the number of live pairs, the dead (t, -1) pair excluded); get_slopes (models/fdm_vocaset.py:96-106);
 * the positional tables (:150-184). */
int fdm_schedule_host(int T, float* out12);
int fdm_ddim_schedule_host(int steps, int T, int* t, int* t_next, float* sqrt_an, float* c_n);
int fdm_alibi_slopes_host(int n_head, float* out);
int fdm_pe_table_host(int d, int periodic, int period, int rows, float* out);

#ifdef __cplusplus
}
#endif
#endif
