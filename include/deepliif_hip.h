/*
 * deepliif_hip.h -- C ABI of libdeepliif_hip.so: the MI355X (gfx950) kernels under the DeepLIIF cGAN hot path.
 *
 * The reference (nadeemlab/DeepLIIF) has no native layer: its hot path is a sequence of torch.nn module calls
 * (SURVEY.md 2.2).  Each entry point below replaces one family of those calls; the citation is the reference
 * call site (paths relative to /root/reference).  Everything is `extern "C"`, plain pointers and sizes, caller-owned
 * device memory, an explicit hipStream_t (passed as void*), no hidden global stream, no allocation, thread-safe
 * (the only process-wide state is the table of runtime switches below, filled once when the library is loaded and read-only afterwards).
 * Return value: 0 = ok, negative = error (message via dl_last_error(), thread-local).
 * Arguments are validated on the host before anything is launched; an empty problem (N = 0 or a zero-sized image) is an error
 * ("empty problem ..."), not a no-op: the reference never produces one and a silent success would hide a caller bug.
 *
 * Data layout (engine-internal; the host side converts at the seam):
 *   activations  NHWC, channel count padded to a power of two >= 8 ("Cp"), element type bf16 (DL_BF16) or fp32 (DL_F32);
 *                a tensor may be a channel slice of a wider buffer: `pstride` = elements between consecutive pixels.
 *   parameters   fp32 in the reference's own layouts (Conv2d: OIHW, ConvTranspose2d: IOHW), flat per optimizer set;
 *                dl_pack_weights() turns them into the K-contiguous bf16 (hi [+ lo]) images the GEMM kernels stream.
 *   precision    DL_PREC_BF16  : one bf16 MFMA pass, fp32 accumulate.
 *                DL_PREC_BF16X3: operands split a = hi + lo (two bf16), three MFMA passes (hi*hi + hi*lo + lo*hi),
 *                                fp32-class accuracy (~2^-16 relative per product); activations are stored fp32.
 *
 * Two libraries, one ABI.  libdeepliif_hip.so uses bfloat16 as its 16-bit type (training and inference).  libdeepliif_hip_f16.so is the SAME sources
 * compiled with -DDL_H16_FP16: every DL_BF16 tensor, every packed weight image and every MFMA operand is IEEE half instead (11 significand bits
 * instead of 8 at the same matrix rate: 7x nearer to the fp32 reference on a generator's forward).  It is for INFERENCE: the gradients of this model sit
 * below half's normal range, so the host side refuses to build a training model on it (deepliif_amd/engine.py, Precision 'fp16').  Same symbols, same
 * structs, same switches; dl_half_format() tells the two apart.
 */
#ifndef DEEPLIIF_HIP_H
#define DEEPLIIF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DL_VERSION 114

enum { DL_F32 = 0, DL_BF16 = 1 };           /* DL_BF16 = "the 16-bit type of this library": bfloat16, or IEEE half in libdeepliif_hip_f16.so (below) */
enum { DL_HALF_BF16 = 0, DL_HALF_FP16 = 1 };   /* dl_half_format() */
enum { DL_PREC_BF16 = 1, DL_PREC_BF16X3 = 3 };
enum { DL_ACT_NONE = 0, DL_ACT_RELU = 1, DL_ACT_LRELU = 2, DL_ACT_TANH = 3,       /* LRELU slope 0.2 (networks.py:578,639) */
       DL_ACT_SIGMOID = 4 };   /* nn.Sigmoid of the attention gate (att_unet.py:100-104): elementwise entry points only (dl_act_forward / _backward) */
enum { DL_PAD_ZERO = 0, DL_PAD_REFLECT = 1 };
enum { DL_NORM_INSTANCE = 0, DL_NORM_BATCH = 1 };
enum { DL_LOSS_BCE_LOGITS = 0, DL_LOSS_MSE = 1, DL_LOSS_SMOOTH_L1 = 2, DL_LOSS_L1 = 3,
       DL_LOSS_LINEAR = 4 };     /* target_const * x: GANLoss('wgangp') = -mean(pred) for real, +mean(pred) for fake (networks.py:307-311) */

#define DL_MAX_TAPS 64
#define DL_MAX_PHASES 4
#define DL_WGRAD_MULTI_MAX 24       /* layers per dl_conv_wgrad_multi launch (their pointers travel as kernel arguments) */

int dl_version(void);
const char *dl_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Runtime switches.  The shipped library looks at NINE environment variables, each a choice between two CORRECT code paths (kept for same-box A/B
 * measurements and for tests that must reach a kernel at small sizes).  They are copied out of the environment ONCE, when the library is loaded; no entry
 * point calls getenv.  Anything that can change RESULTS (timing-only ablations) or selects superseded kernel variants exists only in the dev build
 * (`make -C deepliif_amd/csrc dev` -> libdeepliif_hip_dev.so, compiled with -DDL_DEV_SWITCHES; dl_dev_build() == 1), which tools/ load explicitly.
 *   DL_CONV_S2F=0      stride-2 transposed convs / stride-2 data gradients on the 4-phase gather GEMM instead of conv_s2f_kernel; =2 lifts its size rule
 *   DL_CONV_S2FX3=0    the same for the strict policy's conv_s2f_x3_kernel
 *   DL_CONV_S2D=0      ResnetGenerator down1 forward / up2 data gradient on the gather GEMM instead of conv_s2d_kernel, up2 forward / down1 data
 *                      gradient on conv_s2f_kernel instead of conv_s2u_kernel
 *   DL_CONV_DOT=0      the PatchGAN's one-channel prediction layer (forward / data gradient) and its first layer (forward / data gradient) on the gather GEMM
 *                      instead of conv_dot_*_kernel / conv_d1_kernel / conv_d1g_kernel
 *   DL_CONV_W4X3=1     strict ResnetBlock conv on conv_gemm_w4x3_kernel (opt-in; a measured tie with the default 8-phase strict kernel)
 *   DL_PACK_TILED=0    dl_pack_weights_batch with every image in the chunk-per-thread form
 *   DL_NO_X3_GLDS      (set) strict policy on the register-staged round-1 kernels           -- deepliif_amd/ops.py reads the same variable
 *   DL_NO_WGRAD_C4=1   7x7 stem / head weight gradient on the general kernels               -- deepliif_amd/ops.py reads the same variable
 *   DL_NO_C4_X3        (set) strict 7x7 stem / head on the general strict kernels           -- deepliif_amd/ops.py reads the same variable
 * dl_switches_reload() re-reads the nine variables (tests that flip one inside a process); NOT thread-safe against concurrent launches.
 * ---------------------------------------------------------------------------------------------------------- */
int dl_switch_count(void);
const char *dl_switch_name(int id);          /* 0 <= id < dl_switch_count() */
void dl_switches_reload(void);
int dl_half_format(void);                    /* DL_HALF_BF16: libdeepliif_hip.so; DL_HALF_FP16: libdeepliif_hip_f16.so */
int dl_dev_build(void);                      /* 1: compiled with -DDL_DEV_SWITCHES (A/B variants and timing-only ablations reachable); 0: the shipped library */
/* number of bytes of fp32 scratch a call needs; see each function */

/* ------------------------------------------------------------------------------------------------------------
 * Gather-GEMM: every Conv2d / ConvTranspose2d forward and every data-gradient on the path.
 *   out[n, hq*out_step+oh_p, wq*out_step+ow_p, co] = act( bias[co] +
 *        sum_{t in taps(p)} sum_{ci} in[n, hq*in_step+dh_t, wq*in_step+dw_t, ci] * W[co, kbase_p + (t-t0_p)*Ci + ci] )
 * for every phase p (sub-pixel phases make a stride-2 transposed conv four dense GEMMs, no zero insertion).
 * Replaces: nn.Conv2d / nn.ConvTranspose2d forward in networks.py:386-444 (ResnetGenerator), :490-506 (ResnetBlock),
 * :576-609 (UnetSkipConnectionBlock), :638-660 (NLayerDiscriminator) and ATen's conv backward-data reached from
 * loss_D.backward()/loss_G.backward() (DeepLIIF_model.py:332,429).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct dl_conv_desc {
    int32_t N, Hi, Wi, Ci;          /* input: Ci = padded channels (power of two >= 8)            */
    int32_t in_pstride;             /* elements between consecutive input pixels                   */
    int32_t Ho, Wo, Co;             /* output: Co = channels to write (padded count, multiple of 8) */
    int32_t out_pstride;
    int32_t Hq, Wq;                 /* per-phase iteration grid                                    */
    int32_t out_step, in_step;
    int32_t n_phase;
    int32_t phase_oh[DL_MAX_PHASES], phase_ow[DL_MAX_PHASES];
    int32_t phase_tap_begin[DL_MAX_PHASES + 1];
    int32_t phase_kbase[DL_MAX_PHASES];     /* column offset of the phase in a packed weight row (multiple of 64) */
    int8_t tap_dh[DL_MAX_TAPS], tap_dw[DL_MAX_TAPS];
    int32_t pad_mode;               /* DL_PAD_*: reflect only for single-phase, in_step==1 layers  */
    int32_t w_kstride;              /* packed weight row length (elements)                         */
    int32_t w_rows;                 /* packed weight rows (Co rounded up to 128)                   */
    int32_t act;                    /* epilogue activation DL_ACT_*                                */
    int32_t in_dtype, out_dtype;    /* DL_F32 / DL_BF16                                            */
    int32_t prec;                   /* DL_PREC_*                                                   */
    int32_t splitk;                 /* >1: partial sums go to `slab` (fp32 [splitk][N*Ho*Wo][Co]) and are combined by
                                       the same call in a second, fixed-order kernel (deterministic)              */
    int32_t in_act;                 /* activation applied to the INPUT while it is staged (DL_ACT_NONE/RELU/LRELU):
                                       UnetSkipConnectionBlock's pre-activation (networks.py:578-602)             */
    int32_t bias_n;                 /* valid entries of `bias` (the real channel count; bias may be unaligned)    */
    int32_t raw_out;                /* 1: write the raw fp32 accumulators to `slab` ([N*Ho*Wo][Co]) and nothing to `out`
                                       (first half of the narrow-Cout path, see dl_shift_sum)                            */
    int32_t ci_real;                /* real (unpadded) contracted channels; 0 = unknown.  <= 4 with Ci == 8 lets a stride-1 7x7 layer take the
                                       4-channel patch kernel (csrc/conv_c4.h: ResnetGenerator stem forward, head data gradient)               */
    int32_t in_split;               /* 1 (strict policy only: DL_F32 + DL_PREC_BF16X3, in_act == DL_ACT_NONE): `in` is the SPLIT COPY of the fp32
                                       activations -- same addressing, every group of 8 channels holds [8 bf16 hi | 8 bf16 lo] (written by
                                       dl_norm_forward / dl_norm_backward, z_split / dy_split) -- so the kernels skip their own hi / lo split   */
} dl_conv_desc;

int dl_conv_forward(const dl_conv_desc *d, const void *in, const void *w_hi, const void *w_lo, const float *bias,
                    void *out, float *slab, float *stats_part, void *stream);
/* Fused normalisation statistics: when > 0, dl_conv_forward can also emit the per-(image, channel) partial sums / sums of
 * squares of the values it stores (stats_part = fp32 [N][chunks][2][Co], chunks = the returned value) so that dl_norm_forward
 * (dl_norm_desc.ext_nchunks = chunks, partials at the start of its `ws`) skips its own statistics pass over y.  0 = not
 * available for this descriptor (pass stats_part = NULL). */
int dl_conv_stats_chunks(const dl_conv_desc *d);
/* Data gradient with the FOLLOWING normalisation backward's reductions fused into its store epilogue.  The conv's output here is dz,
 * the gradient with respect to z = act(norm(y)) of the layer in front (networks.py:381-384 Conv2d -> norm -> ReLU; ResnetBlock
 * :478-500; NLayerDiscriminator :644-652); dl_norm_backward needs  S1 = sum dn  and  S2 = sum dn * xhat  over every image,
 * dn = dz * act'(y * scale + shift), xhat = (y - mean) * rstd, before it can apply -- normally its own pass over y and dz.  With
 * `bn` the conv reads the matching y tile next to the dz tile it is about to store and leaves the per-tile partials in
 * stats_part = fp32 [N][chunks][2][Co] (chunks = dl_conv_bnstats_chunks(d) > 0), which dl_norm_backward consumes through
 * dl_norm_desc.ext_nchunks (partials at the start of its `ws`).  Valid only when this conv's output is the ONLY contribution to dz. */
typedef struct dl_conv_bnstats {
    const void *y;                  /* pre-norm tensor, same N x Ho x Wo pixels and dtype as the conv output          */
    int32_t y_pstride;
    int32_t act;                    /* activation behind the norm: DL_ACT_NONE / DL_ACT_RELU / DL_ACT_LRELU             */
    const float *mean, *rstd, *scale, *shift;      /* [N][Co], as dl_norm_forward left them                            */
} dl_conv_bnstats;
int dl_conv_bnstats_chunks(const dl_conv_desc *d);
int dl_conv_forward_bnstats(const dl_conv_desc *d, const void *in, const void *w_hi, const void *w_lo, void *out,
                            float *stats_part, const dl_conv_bnstats *bn, void *stream);
/* out = conv(in) + addend in one pass (bf16; the sum is formed in fp32 before the store's rounding).  Replaces the separate accumulation of a second gradient
 * contribution where the data gradient of a residual block's first conv meets the gradient that came down the skip connection (networks.py:509-513,
 * ResnetBlock.forward: out = x + conv_block(x); autograd adds the two contributions of x).  dl_conv_add_supported: 1 when the kernel dl_conv_forward picks for
 * `d` has the fused form (today the ResnetBlock shape on conv_gemm_w4_kernel), else 0 -- the caller then adds with dl_axpby.  `addend` may alias `out`. */
int dl_conv_add_supported(const dl_conv_desc *d);
int dl_conv_forward_add(const dl_conv_desc *d, const void *in, const void *w_hi, const void *w_lo, const void *addend, int32_t addend_pstride,
                        void *out, void *stream);
/* Name of the kernel dl_conv_forward would launch for `d` (static string, matches the rocprofv3 kernel name up to template
 * arguments).  Diagnostic only: bench.py labels its roofline line with it. */
const char *dl_conv_kernel_name(const dl_conv_desc *d);

/* ------------------------------------------------------------------------------------------------------------
 * Weight gradient:  grad[a, b, kh, kw] (+)= sum_{n,hp,wp} P[n,hp,wp,a] * Q[n, hp*step - pad + kh, wp*step - pad + kw, b]
 *   Conv2d           : P = dL/dy (a = out channel), Q = layer input  (b = in channel)  -> OIHW
 *   ConvTranspose2d  : P = layer input (a = in channel), Q = dL/dy (b = out channel)   -> IOHW
 * (pixel contraction runs on MFMA via ds_read_b64_tr_b16 transposing LDS reads; split-K slabs combined in a fixed order)
 * Replaces ATen conv backward-weight reached from DeepLIIF_model.py:332,429.
 * slab: fp32 scratch, dl_wgrad_slab_floats(d) elements (splitk slabs of CAp * (KH*KW*CBp), padded apart).  reflect padding of Q is supported (pad_mode).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct dl_wgrad_desc {
    int32_t N, Hp, Wp, CAp, p_pstride;   /* P: coarse grid                                   */
    int32_t Hq, Wq, CBp, q_pstride;      /* Q: gathered grid                                 */
    int32_t KH, KW, step, pad, pad_mode;
    int32_t CA, CB;                      /* real channel counts of the fp32 gradient tensor  */
    int32_t dtype;                       /* dtype of P and Q                                 */
    int32_t prec;
    int32_t splitk;
    int32_t accumulate;                  /* 0: grad = result, 1: grad += result              */
    int32_t q_act;                       /* activation applied to Q while staged (see dl_conv_desc.in_act) */
    int32_t p_act;
    int32_t pad_w;                       /* horizontal padding; -1 = same as `pad` */
    int32_t stack_kw;                    /* >0: P's channel a' = a*stack_kw + kw (dl_shift_stack image), KW must be 1; the result goes to
                                            grad[a][b][kh][kw] with stack_kw kernel columns */
    int32_t p_split, q_split;            /* 1: P / Q is the split copy of the fp32 tensor (see dl_conv_desc.in_split); strict policy, no staged activation */
} dl_wgrad_desc;

/* floats the `slab` argument of dl_conv_wgrad / dl_conv_wgrad_slabs must hold for this descriptor: splitk x (CAp x KH*KW*CBp + pad); pad = DL_WGRAD_SLAB_PAD
 * floats (default 0: a padded slab distance was measured to change nothing, see csrc/wgrad.hip). */
size_t dl_wgrad_slab_floats(const dl_wgrad_desc *d);
int dl_conv_wgrad(const dl_wgrad_desc *d, const void *P, const void *Q, float *grad, float *slab, void *stream);

/* Deferred reduction of the split-K slabs (round 4).  A backward pass over one network launches ~30-60 weight gradients; reducing each one's
 * slabs right behind it costs a ~20 us launch per layer that reads what the previous kernel has just written.  With 288 GB of HBM the slabs of a
 * whole network fit side by side, so:
 *   dl_conv_wgrad_slabs     runs ONLY the split-K kernel into `slab` (a region the caller keeps untouched until the batch has run) and fills
 *                           *entry_host (HOST memory) with the record of the pending reduction (block0 = 0, nblocks = its share of the grid);
 *   dl_wgrad_reduce_batch   combines `count` pending reductions in ONE launch; table_dev = the records in DEVICE memory, sorted by block0 with
 *                           block0 = running sum of nblocks (the caller fills it in), total_blocks = the sum.
 * Two records of one batch must not name the same `grad` (their accumulation would race): the caller flushes before a second use.
 * Per element the summation order is dl_conv_wgrad's, so results are bit-identical to the immediate form.
 * dl_conv_wgrad_deferrable: 0 for the persistent narrow-channel forms (7x7 stem / head), which reduce in place -- use dl_conv_wgrad. */
typedef struct dl_wgrad_reduce_entry {
    const float *slab;
    float *grad;
    int32_t splitk, CAp, CBp, J, CA, CB, KK, accumulate, stack_kw;
    int32_t block0, nblocks, kstride;    /* kstride: floats between consecutive slabs (see dl_wgrad_slab_floats) */
} dl_wgrad_reduce_entry;
int dl_conv_wgrad_deferrable(const dl_wgrad_desc *d);
int dl_conv_wgrad_slabs(const dl_wgrad_desc *d, const void *P, const void *Q, float *grad, float *slab, dl_wgrad_reduce_entry *entry_host, void *stream);
int dl_wgrad_reduce_batch(const dl_wgrad_reduce_entry *table_dev, int count, int total_blocks, void *stream);

/* Batched weight gradient (round 5).  With one launch per layer the split-K factor must fill 256 CUs by itself (ResnetBlock conv: 9 tiles x 28
 * pixel ranges -> 28 fp32 partial copies of every gradient, 66 MB written and read back for a 2.4 MB result; every workgroup's prologue and slab
 * store sit exposed).  The weight gradients of a backward pass do not depend on each other, and with 288 GB of HBM the operands (dL/dy, x) of a
 * whole network stay alive until its pass ends, so same-shaped layers are computed by ONE launch: n layers x tiles x splitk workgroups, several
 * rounds of the chip, split-K 3-7 instead of 28, the slab stores of one round hidden behind the next round's main loops.
 *   dl_wgrad_plan        which kernel dl_conv_wgrad would run for `d` (pass splitk = 1): *tiles = its output tiles per layer, *ksteps = K steps of one
 *                        tile at split-K 1, *name = kernel name (static string, diagnostic); returns 1 when that kernel has a batched form, 0 when
 *                        not, < 0 on error.  Callers size split-K from it: one layer: tiles x splitk ~ one round of 256 CUs; a batch: see
 *                        deepliif_amd/geometry.py choose_wgrad_batch_splitk.
 *   dl_conv_wgrad_multi  n <= DL_WGRAD_MULTI_MAX layers sharing the descriptor `d` (HOST arrays P[n], Q[n], grad[n]) in one split-K launch; layer
 *                        l's slabs start at slab + l * dl_wgrad_slab_floats(d); entries_host[l] = its pending reduction for dl_wgrad_reduce_batch
 *                        (as dl_conv_wgrad_slabs).  Per element the result is the fixed-order sum of d->splitk partials: deterministic, and
 *                        bit-identical to dl_conv_wgrad with the same splitk.
 * Replaces the per-layer ATen conv backward-weight calls reached from DeepLIIF_model.py:332,429 (networks.py:467-513 for the dominant shape). */
int dl_wgrad_plan(const dl_wgrad_desc *d, int32_t *tiles, int32_t *ksteps, const char **name);
int dl_conv_wgrad_multi(const dl_wgrad_desc *d, int n, const void *const *P, const void *const *Q, float *const *grad, float *slab,
                        dl_wgrad_reduce_entry *entries_host, void *stream);

/* Pack an fp32 parameter tensor src[A][B][KH][KW] into the K-contiguous bf16 image(s) dl_conv_forward streams.
 *   row_is_a != 0 : packed row = a, contracted channel = b   (Conv2d forward; ConvTranspose2d data-grad)
 *   row_is_a == 0 : packed row = b, contracted channel = a   (ConvTranspose2d forward; Conv2d data-grad)
 * Column layout: for each phase p, taps [tap_begin[p], tap_begin[p+1]) x Cc (padded contracted channels), zero padded up
 * to the next multiple of 64 columns.  tap_kh/tap_kw give each tap's kernel coordinates.  w_lo may be NULL. */
typedef struct dl_pack_desc {
    int32_t A, B, KH, KW;
    int32_t row_is_a;
    int32_t rows_real, rows_pad;         /* rows_pad multiple of 128                       */
    int32_t Cc, Cc_pad;                  /* contracted channels, real / padded (pow2 >= 8) */
    int32_t n_phase;
    int32_t phase_tap_begin[DL_MAX_PHASES + 1];
    int32_t phase_kbase[DL_MAX_PHASES];
    int8_t tap_kh[DL_MAX_TAPS], tap_kw[DL_MAX_TAPS];
    int32_t kstride;
    int32_t stack_kw;                    /* 1: packed row = a*KW + kw (kernel column folded into the row index), taps give kh only */
} dl_pack_desc;

int dl_pack_weights(const dl_pack_desc *d, const float *src, void *w_hi, void *w_lo, void *stream);
/* Batched form for "repack everything after an optimizer step" (hundreds of images: one launch, work split by image size).
 * A job record is an opaque blob of dl_pack_job_bytes() bytes that dl_pack_job_fill() writes into HOST memory from the same
 * arguments dl_pack_weights takes (device pointers are only recorded, nothing is launched).  dl_pack_batch_blocks() turns `count`
 * back-to-back host records into the workgroup table {job, first 16-byte chunk} (int32 pairs; pass NULL to get the entry count).
 * The caller copies both tables to device memory once and calls dl_pack_weights_batch after every weight update; they stay valid
 * as long as the descriptors and the three pointers of each job do.  Results are bit-identical to dl_pack_weights.
 * The table is opaque: images whose phases are whole taps back to back (>= 64 contracted channels, kernels up to 4x4) get one entry per
 * 8-row x 64-channel tile (job | 1 << 30, tile) and are packed through LDS in the master weight's own order; DL_PACK_TILED=0 turns that off. */
size_t dl_pack_job_bytes(void);
int dl_pack_job_fill(const dl_pack_desc *d, const float *src, void *w_hi, void *w_lo, void *job_host);
int dl_pack_batch_blocks(const void *jobs_host, int count, int32_t *block_tab_host);
int dl_pack_weights_batch(const void *jobs_dev, const int32_t *block_tab_dev, int nblocks, void *stream);

/* Narrow-Cout convolutions (the 7x7 ResnetGenerator head, networks.py:438-443: 64 -> 3 channels): an MFMA tile has at least 16
 * output rows, so Cout = 3 would waste 13/16 of the matrix pipe.  The kernel column is folded into the GEMM rows instead:
 *   T[n,h,w,(co,kw)] = sum_{kh,ci} x[n,h+kh-pad,w,ci] * W[co,ci,kh,kw]      (dl_conv_forward, KH vertical taps, Cout*KW rows, raw_out)
 *   y[n,h,w,co]      = act(bias[co] + sum_kw T[n,h,w+kw-pad,co*KW+kw])      (dl_shift_sum)
 * and for the weight gradient  D[n,h,w,(co,kw)] = dy[n,h,w-(kw-pad),co]  (dl_shift_stack) feeds dl_conv_wgrad with KH x 1 taps. */
/* Narrow-Cout ConvTranspose2d(k=4, s=2, p=1) (UnetGenerator's outermost up-convolution to 3 channels + Tanh, networks.py:573-576), second
 * half: T[n,y,x,(ky*4+kx)*Cout+co] is the raw fp32 result of ONE 1x1 GEMM over the input (dl_conv_forward(raw_out) with rows = (ky,kx,co));
 * out[n,oy,ox,co] = act(bias[co] + the 2x2 contributions with oy = 2y-1+ky, ox = 2x-1+kx).  Cout <= 4; out is N x 2H x 2W x oCp. */
int dl_convt4_gather(const float *T, int N, int H, int W, int Tc, int Cout, const float *bias, int act, int out_dtype, void *out,
                     int out_pstride, int out_Cp, void *stream);
int dl_shift_sum(const float *T, int N, int H, int W, int Tc, int Cout, int KW, int pad, int pad_mode, const float *bias, int act,
                 int out_dtype, void *out, int out_pstride, int out_Cp, void *stream);
int dl_shift_stack(int dtype, const void *dy, int dy_pstride, int N, int H, int W, int Cout, int KW, int pad, int pad_mode,
                   void *D, int Dc, void *stream);

/* The same layer (64 -> Cout <= 4 channels, 7 x KW taps, zero padding, bf16) in ONE kernel: every input row is staged once and multiplies the
 * packed weights of all kernel rows into rolling accumulator blocks, the kernel-column sum runs from LDS (csrc/conv_small.hip).  w_hi is the
 * SAME stack_kw image (rows co*KW + kw, columns kh*Ci + ci).  dl_conv_narrow_supported() says whether a layer qualifies; the two-call form
 * above remains for everything else (fp32 policy, reflection padding, other shapes). */
int dl_conv_narrow_supported(int dtype, int Ci, int x_pstride, int Cout, int KH, int KW, int pad, int pad_mode);
int dl_conv_narrow_forward(const void *x, int N, int H, int W, int Ci, int x_pstride, const void *w_hi, int w_kstride, int Cout, int KH, int KW,
                           int pad, const float *bias, int act, void *out, int out_pstride, int out_Cp, void *stream);
/* The same layer under the strict policy (fp32 storage, split-bf16 x3 products): x and out are fp32 NHWC, w_hi / w_lo the two images of
 * dl_pack_weights(stack_kw).  dl_conv_narrow_supported(DL_F32, ...) says whether it applies.  (round 4; the round-3 route was
 * dl_conv_forward(raw_out) + dl_shift_sum.) */
int dl_conv_narrow_forward_x3(const void *x, int N, int H, int W, int Ci, int x_pstride, const void *w_hi, const void *w_lo, int w_kstride,
                              int Cout, int KH, int KW, int pad, const float *bias, int act, void *out, int out_pstride, int out_Cp, void *stream);

/* Backward of nn.ReflectionPad2d(pad) in front of a padding=0 Conv2d (ResnetGenerator with padding_type='reflect':
 * networks.py:386-388 stem, 478-481 / 495-498 ResnetBlock, 438-440 head).  The data gradient of such a layer is computed in two
 * steps: dl_conv_forward with the pad-0 data-gradient plan gives the gradient with respect to the explicitly padded input
 * (src: [N, H+2*pad, W+2*pad, Cp]); dl_reflect_fold adds every mirrored border row / column back onto the interior pixel it
 * was copied from (dst: [N, H, W, Cp]).  pad < min(H, W) as for nn.ReflectionPad2d.  fp32 accumulation, fixed order. */
int dl_reflect_fold(int dtype, const void *src, int src_pstride, void *dst, int dst_pstride, int N, int H, int W, int pad, int Cp,
                    void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Normalisation (networks.py:25-44: BatchNorm2d on batch statistics / InstanceNorm2d; eps 1e-5, biased variance)
 * fused with the activation that follows it and the ResnetBlock residual add (networks.py:512).
 *   stats    : per-(n,c) sum / sum of squares partials                      -> ws
 *   finalize : mean/rstd per (n,c) [instance] or per c [batch]; scale = gamma*rstd, shift = beta - mean*scale;
 *              optional BatchNorm running-stat update (momentum 0.1, unbiased variance)
 *   apply    : z = act(y*scale + shift) (+ residual)
 * stat buffers: float [N][Cp] each (batch scope replicates the per-channel value over n).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct dl_norm_desc {
    int32_t N, H, W, Cp, C;          /* Cp padded channels, C real                        */
    int32_t y_pstride, z_pstride, r_pstride;
    int32_t dtype;
    int32_t scope;                   /* DL_NORM_*                                         */
    int32_t act;
    float eps;
    float momentum;                  /* <0: do not touch running stats                    */
    int32_t ext_nchunks;             /* >0: `ws` already starts with [N][ext_nchunks][2][Cp] partial sums: forward -- sum y, sum y^2
                                        (dl_conv_forward); backward -- sum dn, sum dn*xhat (dl_conv_forward_bnstats)                  */
} dl_norm_desc;

size_t dl_norm_ws_floats(const dl_norm_desc *d);     /* scratch needed by forward and backward (floats) */

int dl_norm_forward(const dl_norm_desc *d, const void *y, const float *gamma, const float *beta,
                    float *running_mean, float *running_var,
                    float *mean, float *rstd, float *scale, float *shift,
                    const void *residual, void *z, float *ws, void *z_split, void *stream);
/* dl_norm_forward: z / dl_norm_backward: dy may be NULL when z_split / dy_split is given (every consumer of the tensor reads the split copy: one 4-byte
 * store per element less).
 * z_split / dy_split (may be NULL; DL_F32 only): a dense [N][H][W][Cp] buffer of the SAME byte size as the fp32 tensor that receives the split
 * copy of z / dy -- per group of 8 channels 16 bytes of bf16 hi = bf16(v) followed by 16 bytes of bf16 lo = bf16(v - hi) -- for the strict-policy
 * convolutions that consume the tensor next (dl_conv_desc.in_split, dl_wgrad_desc.p_split / q_split): the hi / lo split a conv kernel would redo
 * for every tap and in every consuming wave is done once, by the producer, at the cost of one more 4-byte-per-element store. */

/* dy = d/dy of [ z = act(norm(y)) (+res) ] given dz; dgamma/dbeta (+)= ...; the residual branch receives dz itself.
 * Strides: y uses d->y_pstride, dz uses d->z_pstride, dy uses d->r_pstride.
 * dy_chansum (may be NULL): dy_chansum[c] += sum over pixels of dy[.,c] -- the gradient of the bias of the convolution that
 * produced y (Conv2d(bias=True) in front of InstanceNorm2d, networks.py:381-384), fused into the same pass. */
int dl_norm_backward(const dl_norm_desc *d, const void *dz, const void *y, const float *gamma,
                     const float *mean, const float *rstd, const float *scale, const float *shift,
                     void *dy, float *dgamma, float *dbeta, int accumulate_affine, float *dy_chansum, float *ws, void *dy_split, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Elementwise (networks.py nn.ReLU / nn.LeakyReLU(0.2) / nn.Tanh, torch.cat, the seg weighted sum
 * DeepLIIF_model.py:203, gradient accumulation).  All take NHWC views: (ptr, pstride, npix, Cp).
 * ---------------------------------------------------------------------------------------------------------- */
int dl_act_forward(int act, int dtype, const void *x, int x_pstride, void *y, int y_pstride, int64_t npix, int Cp, void *stream);
/* dx = dy * act'(.) evaluated from the activation OUTPUT y (relu/lrelu: sign, tanh: 1-y^2) */
int dl_act_backward(int act, int dtype, const void *dy, int dy_pstride, const void *y, int y_pstride,
                    void *dx, int dx_pstride, int64_t npix, int Cp, void *stream);
/* Attention gate of the attention U-Net (att_unet.py:84-115, `unet_512_attention`, networks.py:189-190): Attention_block.forward returns
 * x * psi with psi = Sigmoid(BatchNorm2d(Conv2d(F_int, 1, 1)(relu(W_g(g) + W_x(x))))) -- ONE channel broadcast over the C channels of x.
 *   forward : out[p][c] = x[p][c] * psi[p][0]                                 (psi: NHWC with >= 8 padded channels, channel 0 real)
 *   backward: dx[p][c] = g[p][c] * psi[p][0];  dpsi[p][0] = sum_c g[p][c] * x[p][c],  dpsi[p][1..7] = 0    (dx may be NULL) */
int dl_gate_forward(int dtype, const void *x, int x_pstride, const void *psi, int psi_pstride, void *out, int out_pstride, int64_t npix, int Cp,
                    void *stream);
int dl_gate_backward(int dtype, const void *g, int g_pstride, const void *x, int x_pstride, const void *psi, int psi_pstride, void *dx,
                     int dx_pstride, void *dpsi, int dpsi_pstride, int64_t npix, int Cp, void *stream);
/* nn.Dropout(p) in training mode (networks.py:493-494, 604-605): y = x * keep / (1-p), keep = hash(seed, element) >= p.
 * Calling it again with the same seed on the gradient reproduces the same mask (backward); y may alias x. */
int dl_dropout(int dtype, const void *x, int x_pstride, void *y, int y_pstride, int64_t npix, int Cp, float p, uint64_t seed, void *stream);
/* out = alpha*a + beta*b   (b may be NULL; out may alias a or b) */
int dl_axpby(int dtype, float alpha, const void *a, int a_pstride, float beta, const void *b, int b_pstride,
             void *out, int out_pstride, int64_t npix, int Cp, void *stream);
/* copy `C` channels of src (starting at channel src_c0) into dst starting at channel dst_c0 (any C, scalar path) */
int dl_copy_channels(int dtype, const void *src, int src_pstride, int src_c0, void *dst, int dst_pstride, int dst_c0,
                     int64_t npix, int C, int accumulate, void *stream);
/* per-channel sum over pixels of an NHWC tensor (conv bias gradient): out[c] (+)= sum_p x[p][c], c < C */
int dl_channel_sum(int dtype, const void *x, int pstride, int64_t npix, int Cp, int C, float *out, int accumulate,
                   float *ws, void *stream);
/* boundary layout converters: NCHW fp32 (the reference's tensors) <-> NHWC padded engine tensors */
int dl_nchw_to_nhwc(const float *src, int N, int C, int H, int W, int dtype, void *dst, int dst_pstride, int dst_c0,
                    int zero_pad_to, void *stream);
int dl_nhwc_to_nchw(int dtype, const void *src, int src_pstride, int src_c0, float *dst, int N, int C, int H, int W,
                    void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Losses (networks.py:244-317 GANLoss 'vanilla' = BCEWithLogits(mean) / 'lsgan' = MSE(mean) against a constant
 * target; DeepLIIF_model.py:123 SmoothL1Loss(beta=1, mean)).  Over the C real channels of an NHWC tensor.
 *   loss_out[0] = mean loss (fp32, stays on the device).  grad (may be NULL) = grad_scale * dloss/dx.
 *   target: constant `target_const` when `target` is NULL, else a tensor like x.
 * ws: fp32 scratch of dl_loss_ws_floats() floats.
 * ---------------------------------------------------------------------------------------------------------- */
size_t dl_loss_ws_floats(void);
int dl_loss(int kind, int dtype, const void *x, int x_pstride, const void *target, int t_pstride, float target_const,
            int64_t npix, int C, int Cp, float *loss_out, void *grad, int g_pstride, float grad_scale,
            float *ws, void *stream);
/* same, with loss_out[0] = (accumulate ? loss_out[0] : 0) + out_scale * mean loss: the five weighted nn.L1Loss terms of VGGLoss
 * (networks.py:732-743, DL_LOSS_L1) sum into one slot without a host round trip */
int dl_loss_acc(int kind, int dtype, const void *x, int x_pstride, const void *target, int t_pstride, float target_const,
                int64_t npix, int C, int Cp, float *loss_out, float out_scale, int accumulate, void *grad, int g_pstride, float grad_scale,
                float *ws, void *stream);

/* nn.Upsample(scale_factor=2, mode='nearest') of ResnetGenerator's --upsample resize_conv route (networks.py:409-415).
 * backward == 0: src [N, H, W, Cp] -> dst [N, 2H, 2W, Cp];  backward != 0: src = dL/dy [N, 2H, 2W, Cp] -> dst = dL/dx [N, H, W, Cp] (2 x 2 block sums). */
int dl_upsample2_nearest(int dtype, int backward, const void *src, int src_pstride, void *dst, int dst_pstride, int N, int H, int W, int Cp, void *stream);

/* DeepLIIFKD's distillation term (DeepLIIFKD_model.py:313-336): KLDivLoss(reduction='batchmean') between LogSoftmax(x.view(1, 1, -1)) and
 * Softmax(t.view(1, 1, -1)) -- ONE softmax over all npix * C real elements of each tensor:  KL = sum_j p_j (log p_j - log q_j), p = softmax(t),
 * q = softmax(x).  loss_out[0] = (accumulate ? loss_out[0] : 0) + out_scale * KL;  grad (may be NULL) = grad_scale * (q - p), padded channels 0.
 * Inputs are generator outputs (|x|, |t| <= 1): the exponentials are taken without a running maximum.  ws: dl_kldiv_ws_floats() floats. */
size_t dl_kldiv_ws_floats(void);
int dl_kldiv(int dtype, const void *x, int x_pstride, const void *t, int t_pstride, int64_t npix, int C, int Cp,
             float *loss_out, float out_scale, int accumulate, void *grad, int g_pstride, float grad_scale, float *ws, void *stream);

/* nn.MaxPool2d(kernel_size=2, stride=2) of torchvision's VGG19 `features` (networks.py:698-731 slices them): y [N, H/2, W/2, Cp];
 * backward routes dy to the first maximum of each window in row-major window order (ATen's tie rule) and zeroes everything else. */
int dl_maxpool2_forward(int dtype, const void *x, int x_pstride, void *y, int y_pstride, int N, int H, int W, int Cp, void *stream);
int dl_maxpool2_backward(int dtype, const void *x, int x_pstride, const void *dy, int dy_pstride, void *dx, int dx_pstride,
                         int N, int H, int W, int Cp, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Adam over one flat fp32 parameter set (torch.optim.Adam semantics: networks.py:46-53, DeepLIIF_model.py:128-147;
 * eps 1e-8, no weight decay, bias correction by step count).  grad_scale multiplies the gradient first (1/world_size
 * after a sum all-reduce).
 * ---------------------------------------------------------------------------------------------------------- */
int dl_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                 float lr, float beta1, float beta2, float eps, int step, float grad_scale, void *stream);
/* The same step with its per-step scalars in device memory, for a captured hipGraph of optimize_parameters() (kernel arguments are frozen at capture;
 * learning rate and bias corrections are not): dl_adam_hyper fills hyper_host[8] = {lr, beta1, beta2, eps, 1 - beta1^step, sqrt(1 - beta2^step),
 * grad_scale, 0} with the expressions dl_adam_step uses; the caller copies it to hyper_dev before every replay.  Bit-identical to dl_adam_step. */
int dl_adam_hyper(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float *hyper_host);
int dl_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const float *hyper_dev, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Tiles: the crop / is_empty / stitch steps either side of the generator DAG, on uint8 RGB images resident in HBM
 * ([H][W][3] bytes, `row_stride` bytes between rows).  Integer and byte work, bit-exact with the reference:
 *   dl_tile_gather_u8     InferenceTiler.__iter__ crop (deepliif/util/__init__.py:258-270, incl. the mirror extension of images smaller
 *                         than a patch :196-211 and the solid border `pad` :268-269) fused with transform()
 *                         (deepliif/data/__init__.py:133-138: ToTensor + Normalize(0.5, 0.5); `lut[v]` = float32((v/255 - 0.5)/0.5), 256
 *                         entries supplied by the caller) -> an engine tile batch [n_tiles][tile][tile][Cp]; up to DL_TILE_MAX_SRC source
 *                         images are concatenated on the channel axis (multi-input models, deepliif/models/__init__.py:276-279).
 *                         origins: device int32 [n_tiles][2] = (x, y) of each tile in the (mirror-extended) image; H0, W0 = the image's
 *                         real size.
 *   dl_tile_gray_stats_u8 is_empty()'s statistic (deepliif/models/__init__.py:391-396 -> util/__init__.py:478-486 image_variance_gray
 *                         -> PIL convert('L')): stats[t] = {count, sum, sum of squares} of the gray values in 1..254 of tile t (uint64);
 *                         variance < 9  <=>  count == 0 or count*sumsq - sum^2 < 9*count^2, evaluated by the caller in integers.
 *   dl_tile_paste_u8      tensor2im() (deepliif/util/util.py:117-135: (x + 1) / 2 * 255 in fp32, truncated to uint8) fused with
 *                         InferenceTiler.stitch (util/__init__.py:272-320): rects = device int32 [n_rects][8] =
 *                         {slot, l, t, w, h, px, py, rgb}: copy the w x h window at (l, t) of tile `slot` of the batch to (px, py) of the
 *                         result image; slot < 0 pastes the constant colour rgb (r | g<<8 | b<<16) instead (empty tiles,
 *                         deepliif/models/__init__.py:399-440).  The caller resolves overlapping pastes (last writer wins) into
 *                         disjoint rectangles, so the launch is order-independent.
 * ---------------------------------------------------------------------------------------------------------- */
#define DL_TILE_MAX_SRC 4
int dl_tile_gather_u8(const void *const *imgs /*host array of device pointers*/, const int64_t *row_strides /*host*/, int n_src, int H0, int W0,
                      const int32_t *origins, int n_tiles, int tile, int pad, uint32_t pad_rgb, const float *lut, int out_dtype, void *out,
                      int out_pstride, int out_cp, void *stream);
int dl_tile_gray_stats_u8(const void *img, int64_t row_stride, int H0, int W0, const int32_t *origins, int n_tiles, int tile, int pad,
                          uint32_t pad_rgb, uint64_t *stats, void *stream);
int dl_tile_paste_u8(int in_dtype, const void *tiles, int in_pstride, int tile, const int32_t *rects, int n_rects, void *dst,
                     int64_t dst_row_stride, void *stream);

/* hardware probes used by the GPU test-suite (MFMA fragment layouts, ds_read_b64_tr_b16 semantics) */
int dl_probe_mfma16(const uint16_t *a /*16x32 bf16 row-major*/, const uint16_t *b /*32x16*/, float *d /*16x16*/, void *stream);
int dl_probe_trread(const uint16_t *src /*64 rows x 16 cols bf16*/, uint16_t *dst /*64 lanes x 4*/, void *stream);
/* Sustained matrix-core rate of THIS box on caller-supplied operand bits (bench.py: roofline.sustained): `blocks` workgroups of four waves, each
 * running iters x 64 register-resident v_mfma_f32_32x32x16_bf16 (the register set of conv_gemm_w4_kernel) on 16 fragments read once from `data`
 * (dl_probe_mfma_sustained_elems(blocks) bf16 values).  flops = blocks * 4 * iters * 64 * 32768.  No reference counterpart: measurement only. */
size_t dl_probe_mfma_sustained_elems(int blocks);
int dl_probe_mfma_sustained(const void *data, int blocks, int iters, float *sink, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Segmentation post-processing (deepliif/postprocessing.py: get_cells_info :311-362 = create_posneg_mask :163-190 + mark_background
 * :193-232 + compute_cell_mapping :235-308; create_cell_classification :923-1000, enlarge_cell_boundaries :1003-1030,
 * create_final_images :1033-1071; driven by compute_final_results :1223-1304).  uint8 H x W x 3 images with a row stride in bytes;
 * bit-exact with the reference (integer work): connected components by union-find, "first in raster order" rules as index minima.
 *
 * dl_pp_cells: seg (+ marker, or the original image together with od_lut[256] = log10(255/i) for the optical-density variant) ->
 *   mask  uint8 [H][W]: BACKGROUND 0 / CELL 100 (the reference's labels after the cell mapping),
 *   label int32 [H][W]: smallest pixel index of the pixel's cell (-1 on background) = the cell's first pixel in raster order,
 *   cells int64 [n][8] in list order: {size, positive pixels, negative pixels, marker max (or optical-density sum), first x, first y,
 *         sum of x, sum of y} for EVERY component (the host applies the noise thresholds and rounds the centroid), at most max_cells rows,
 *   n_cells (device int): number of components, hist (device u64[256], may be NULL): histogram of the gray marker values.
 * dl_pp_finish: code[n_cells] (0 = not counted, 1 = positive, 2 = negative, in the order of `cells`) -> final label mask (in place),
 *   overlay and refined images.  `ws` (dl_pp_ws_bytes) must be the SAME buffer in both calls.
 * ---------------------------------------------------------------------------------------------------------- */
size_t dl_pp_ws_bytes(int H, int W);
int dl_pp_cells(const void *seg, size_t seg_row_stride, const void *marker, size_t marker_row_stride, const double *od_lut,
                int H, int W, int seg_thresh, void *mask, int *label, void *ws, long long *cells, int max_cells,
                int *n_cells, unsigned long long *hist, void *stream);
/* Host-only (no GPU work): the 500-bin Gaussian KDE of calculate_default_size_threshold (postprocessing.py:365-447) over values =
 * sqrt(cell size), in the reference's float64 summation order; returns the index of the first local minimum (1 if none), the bin step
 * through *step, optionally the float32 KDE itself through kde (count entries, may be NULL). */
int dl_pp_kde_first_minimum(const double *values, int n, int count, double *step, float *kde);
int dl_pp_finish(const void *orig, size_t orig_row_stride, void *mask, const int *label, const void *ws, const void *code, int n_cells,
                 int H, int W, void *overlay, size_t overlay_row_stride, void *refined, size_t refined_row_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif
