/*
 * sisr_hip.h -- C ABI of libsisr_hip.so: the MI355X (gfx950) kernels behind the SRGAN hot path
 * of keyber/Single-Image-Super-Resolution.
 *
 * The reference has no FFI of its own: its boundary is the Python nn.Module API
 * (model_generator.py:22-141, model_discriminator.py:18-76, model_content_extractor.py:33-60,
 * utils.py:16-31) and every FLOP runs inside torch.nn primitives.  Each entry point below
 * replaces the torch primitive call sites listed next to it; the Python host side
 * (single-image-super-resolution_amd/) binds them with ctypes and re-creates the reference's
 * module interface on top.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless named host_*;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), allocates nothing,
 *     and returns 0 on success or a hipError_t / negative SISR_E* code; it never aborts;
 *   - activations are NHWC fp32 inside the path; NCHW (the reference's layout) is accepted on the
 *     3-channel image side and produced on the RGB / feature-tap side by layout flags;
 *   - descriptors are filled by the caller for geometry, then completed by the matching
 *     *_plan() call which chooses the tiling and reports workspace sizes.
 */
#ifndef SISR_HIP_H
#define SISR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SISR_E_BADARG (-1)
#define SISR_E_TOOBIG (-2)
#define SISR_E_UNSUPPORTED (-3)

/* ---- input-operand ("prologue") modes: the value fed to the contraction for input element
 *      (n,iy,ix,c); x1/x2 share shape and layout; out-of-image taps contribute 0. ------------- */
enum {
    SISR_PRO_NONE = 0,        /* v = x1                                                        */
    SISR_PRO_ACT = 1,         /* v = lrelu(x1, slope)            PReLU / LeakyReLU / ReLU       */
    SISR_PRO_AFFINE_ACT = 2,  /* v = lrelu(pa[c]*x1 + pd[c], slope)   BatchNorm apply + act     */
    SISR_PRO_BNBWD = 3,       /* v = pa[c]*x1 + pb[c]*x2 + pd[c]      BatchNorm backward        */
    SISR_PRO_BNACT_BWD = 4,   /* z = ps[c]*x2+pt[c]; g = z>0 ? x1 : slope*x1;
                                 v = pa[c]*g + pb[c]*x2 + pd[c]       act' then BatchNorm bwd   */
    SISR_PRO_ACT_BWD = 5,     /* v = x2>0 ? x1 : slope*x1             act' (x2 = pre-activation)*/
    SISR_PRO_TANH_BWD = 6,    /* v = x1*(1 - x2*x2)                   tanh' (x2 = tanh output)  */
    SISR_PRO_RES_AFFINE = 7   /* v = lrelu(x1, slope) + (pa[c]*x2 + pd[c])   residual-block skip sum x + BN2(c2)
                                 (model_generator.py:19) formed in the consuming conv's staging; the sum is also
                                 stored to x_out once per pixel.  Persistent trunk kernels only (forward role):
                                 the generic kernels return SISR_E_UNSUPPORTED. */
};
enum { SISR_X_NHWC = 0, SISR_X_NCHW = 1, SISR_X_NHWC_UNSHUFFLE2 = 2 };
enum { SISR_Y_NHWC = 0, SISR_Y_NCHW = 1, SISR_Y_NHWC_SHUFFLE2 = 2 };
enum { SISR_EPI_NONE = 0, SISR_EPI_TANH = 1 };

/* Packed-weight layout shared by conv / wgrad / weight kernels:
 *   wpk[chunk][r][co][krow], krow = s*PS + (ci - chunk*CK), row length KROWP (zero padded),
 *   co < CoutPad (zero padded).  PS = CK|1 keeps LDS reads conflict-free. */
typedef struct SisrConvPlan {
    int32_t TH, TW, TN;             /* output tile: TN images x TH x TW pixels               */
    int32_t tiles_y, tiles_x, n_groups, n_tiles;
    int32_t CK, PS, KROWP, n_chunk, CoutPad;
    int32_t msub, nsub;             /* 32x32 MFMA sub-tiles per wave (M) / per block (N)     */
    int32_t lds_bytes;
    int32_t wpk_elems;              /* elements in the packed weight buffer                  */
    int32_t variant;                /* bf16 family, bit 0: the weight buffer carries the LANE-ORDER image of the persistent trunk
                                       kernels behind the standard one (SisrWeightDesc.bf_f_lanes / bf_d_lanes); fp32 family, bits 1-2:
                                       the LDS-order image behind the standard one (SisrWeightDesc.f_ldsimg / d_ldsimg: 2 = fp32
                                       values, 4 = split pairs; used only when it matches mfma_split); set by the caller on the
                                       descriptor it launches, 0 after planning                                  */
    /* bf16 family: reciprocals m = ceil(2^32 / d) (0 for d = 1) so that n / d = umulhi(n, m) for n, d < 2^16 --
     * the kernels' index arithmetic (tile id, tile row, LDS row) without integer division */
    uint32_t m_tiles_x, m_thw, m_tw, m_iw, m_wrow;
} SisrConvPlan;

/* Plan of the split-K implicit-GEMM kernels of conv_deep.hip (bf16 build; filled by sisr_conv2d_deep_plan): the layers whose
 * contraction is deep and whose images are small -- the discriminator's strided-conv stack (model_discriminator.py:10,39-44), the
 * VGG19 feature convs (model_content_extractor.py:43) and the generator's trunk at sizes the persistent trunk kernels do not take.
 * A workgroup owns 128 output pixels x BN output channels x one K slice (a range of 32-channel chunks); an output tile is TH rows of
 * the FLATTENED (image, output row) space x TW columns, so that a band of full-width rows may straddle images (12 x 12 and 6 x 6 maps
 * fill 120 / 126 of the 128 MFMA rows); with split > 1 every workgroup leaves its fp32 partial tile in `ws` and
 * conv_deep_finish_kernel sums the slices in a fixed order and runs the epilogue (bias, statistics, reductions, bf16 store). */
typedef struct SisrDeepPlan {
    int32_t enabled;                /* 1: sisr_conv2d_bf16 runs this descriptor on conv_deep.hip            */
    int32_t TH, TW;                 /* output tile: TH flattened output rows x TW columns (<= 128 pixels)   */
    int32_t tiles_x, tiles_q;       /* column tiles, row-band tiles                                         */
    int32_t BN, n_ntiles;           /* output channels per workgroup (64 | 128), cout tiles                 */
    int32_t n_chunk, split, cps;    /* 32-channel chunks, K slices, chunks per slice                        */
    int32_t PR;                     /* padded input rows per image (pad_y + H + bottom padding)             */
    int32_t IW, IH_max;             /* halo tile: columns, most rows of any tile                            */
    int32_t NIT;                    /* 16-byte staging items per thread                                     */
    int32_t wimg_elems;             /* bf16 elements of the weight image [chunk][tap row][Cout][KW*32 + 8]  */
    int32_t classes;                /* 1, or 4: the output-parity classes of a stride-2 data gradient in ONE launch   */
    int64_t ws_bytes;               /* bytes of the split workspace (0 when split == 1)                     */
    uint32_t m_tiles_x, m_tw, m_ho, m_pr, m_iw;
    uint32_t rsv2;
} SisrDeepPlan;

/* Direct convolution, fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32) accumulate.
 * Replaces nn.Conv2d forward and its data-gradient (model_generator.py:10,13,33,39,45,52,123;
 * model_discriminator.py:10,39; torchvision VGG19 convs via model_content_extractor.py:43) with
 * the neighbouring BatchNorm2d-apply / PReLU / LeakyReLU / ReLU (prologue), bias, Tanh,
 * PixelShuffle(2) and residual add (epilogue) and the BatchNorm batch statistics fused in. */
typedef struct SisrConvDesc {
    const float *x1, *x2;                    /* input operand(s)                              */
    const float *pa, *pb, *pd, *ps, *pt;     /* per-input-channel prologue constants          */
    const float *wpk;                        /* packed weights (see above)                    */
    const float *bias;                       /* [Cout] in ORIGINAL channel order, or NULL     */
    const float *res;                        /* residual, same layout as y, or NULL           */
    float *y;
    float *stat_part;                        /* [n_tiles][2][Cout] (mean, M2) or NULL         */
    float *cnt_part;                         /* [n_tiles] valid pixels per tile               */
    /* bf16 kernels, NHWC output: the output y is the gradient g arriving at a BatchNorm (optionally through the
     * leaky activation after it); the epilogue then also emits that BatchNorm's backward reductions per tile --
     * bnb_part[tile][0..C) = sum gg, [C..2C) = sum gg*xhat, [2C] = sum_{z<=0} g*z  (gg = act'(z)*g,
     * z = scale*x + shift, xhat = (x - mean)*invstd) -- in the row format sisr_bn_bwd_finalize reads.  NULL: off. */
    const float *bnb_x, *bnb_scale, *bnb_shift, *bnb_mean, *bnb_invstd;
    const float *bnb_slope_p;
    float *bnb_part;
    float *x_out;                            /* SISR_PRO_RES_AFFINE: the materialised operand, layout / type of x1 */
    /* Deferred BatchNorm finalisation (persistent forward trunk kernels, prologues AFFINE_ACT / RES_AFFINE): when
     * fin_stat is set, pa / pd are NOT read -- every workgroup merges the fin_rows statistics rows (fin_stat [rows][2][Cin]
     * mean / M2, fin_cnt [rows]) of the BatchNorm whose apply the prologue is, exactly as sisr_bn_finalize does (Chan's
     * formula in double), and uses scale = gamma * invstd, shift = beta - mean * scale; workgroup 0 also writes
     * fin_k [4][Cin] = scale, shift, mean, invstd and updates the running statistics (momentum, unbiased variance).
     * Saves the 5-6 us sisr_bn_finalize launch between two convs. */
    const float *fin_stat, *fin_cnt, *fin_gamma, *fin_beta;
    float *fin_rm, *fin_rv, *fin_k;
    int32_t N, H, W, Cin;                    /* logical input                                 */
    int32_t Ho, Wo, Cout;                    /* logical output grid                           */
    int32_t KH, KW, stride, pad_y, pad_x;
    int32_t x_mode, pro_mode;
    const float *pro_slope_p;                /* device scalar slope (PReLU weight); NULL: pro_slope */
    float pro_slope;
    int32_t y_mode, epi_act;
    int32_t bnb_act; float bnb_slope;        /* activation after that BatchNorm: 0 none, 1 leaky (slope / *slope_p) */
    int32_t y_sy, y_oy, y_sx, y_ox, y_H, y_W; /* output pixel (oy*y_sy+y_oy, ox*y_sx+y_ox) of a
                                                y_H x y_W image (strided scatter for the data
                                                gradient of stride-2 convs); plain: 1,0,1,0,Ho,Wo */
    /* storage type of the tensors behind the float* fields (0: fp32, 1: bf16 -- the bf16 build keeps NHWC
     * activations and gradients as bf16 in HBM; all arithmetic, accumulation and statistics stay fp32).
     * x_bf16: x1 and x2;  y_bf16: y (NHWC / NHWC_SHUFFLE2 only);  res_bf16: res;  bnbx_bf16: bnb_x. */
    int32_t x_bf16, y_bf16, res_bf16, bnbx_bf16;
    int32_t fin_rows; float fin_momentum, fin_eps;
    /* fp32 tensors, trunk geometry only: 1 = the contraction runs on the bf16 matrix instruction over (hi, lo) bf16 pairs of every
     * fp32 operand (hi*hi + hi*lo + lo*hi, fp32 accumulate; operands good to 2^-17 relative); 0 = exact fp32 matrix instruction */
    int32_t mfma_split;
    SisrConvPlan plan;
    /* conv_deep.hip (see SisrDeepPlan): `wdeep` = the weight image in that family's layout (SisrWeightDesc.wdp_fwd / wdp_dgrad),
     * `deep_ws` = the caller's split workspace (deep.ws_bytes; may be NULL when deep.split == 1), `epi_scale_p` = device scalar the
     * accumulators are multiplied by before the bias (1 / sigma of a spectrally normalised weight packed un-normalised) or NULL */
    const void *wdeep;
    float *deep_ws;
    const float *epi_scale_p;
    /* deep.classes == 4 -- the data gradient of a stride-2 3x3 convolution (even H, W) as ONE launch over its four output-parity
     * classes c = 2 py + px: the descriptor describes the class convolution with the MOST taps (KH = KW = 2, stride 1, pad 0, over
     * dy; Ho x Wo = the class grid H/2 x W/2; y_sy = y_sx = 2, y_H x y_W = the gradient's size); class c has deep_ckh[c] tap rows,
     * writes output pixel (2 a + py, 2 b + px) and reads its weights from wdeep_c[c] -- always in the KW = 2 row format, a class
     * with one tap per row carries zeros for the second (SisrWeightDesc.wdp_cls_kw = 2).  Partial rows (bnb_part) and split
     * workspace tiles are class-major. */
    const void *wdeep_c[4];
    int32_t deep_ckh[4];
    SisrDeepPlan deep;
} SisrConvDesc;

int sisr_conv2d_plan(SisrConvDesc *d);                        /* host only: fills d->plan     */
int sisr_conv2d_f32(const SisrConvDesc *d, void *stream);

/* Weight gradient of the same convolution: dW[r][s][ci][co] = sum_pix in(pix+tap)[ci]*dy(pix)[co].
 * Replaces the weight/bias part of convolution_backward for the call sites above.  The input
 * operand (x*, p*, x_mode, pro_*) and the output-gradient operand (g*, q*, g_mode, gpro_*) take
 * the same prologue modes.  Each block writes one fp32 partial slab; sisr_wgrad_reduce sums them. */
/* Plan of wgrad_deep.hip (bf16 build, filled by sisr_wgrad_deep_plan after sisr_wgrad_plan_bf16): the weight gradient of the 3x3
 * layers with channels in 64s (the discriminator's conv stack, the generator's trunk at sizes the persistent kernel does not take).
 * A workgroup owns 64 input channels x 64 output channels x all 9 taps (nine 32 x 32 accumulators per wave: the whole block of the
 * gradient stays in registers over all of the workgroup's pixel tiles) and a share of the pixel tiles; the tiles are conv_deep.hip's
 * (rows of the flattened (image, row) space, so 6 x 6 .. 24 x 24 maps waste no MFMA rows), the contraction runs over the tile's
 * positions in HALO coordinates -- dy is laid out on the pitch of the x halo, so that tap (ky, kx) is the x image shifted by a
 * constant -- and both operands are staged one tile ahead. */
typedef struct SisrWgradDeepPlan {
    int32_t enabled;
    int32_t TH, TW, tiles_x, tiles_q, n_tiles;
    int32_t PR;                     /* padded x rows per image                                                   */
    int32_t IW, IH_max;             /* x halo: pitch (even for stride 2), most rows of any tile                  */
    int32_t IWd, IHd_max;           /* dy image in halo coordinates: pitch (IW, or IW / 2 for stride 2), rows    */
    int32_t NPOS_max;               /* positions of the contraction per tile, a multiple of 16                   */
    int32_t XP_max;                 /* x halo pixels incl. the slack the last K step reads                       */
    int32_t NITX, NITD;             /* 16-byte staging items per thread (x, dy)                                  */
    int32_t n_pb, tiles_per_pb;     /* pixel blocks (= slabs) and tiles per block                                */
    int32_t n_cib, n_cob;           /* 64-channel blocks of Cin / Cout                                           */
    int32_t lds_bytes;
    int32_t slab_bf16;              /* the gradient part of a slab row is bf16 (sisr_wgrad_bf16_slab_lead)       */
    int32_t batch_first_wg;         /* sisr_wgrad_deep_batch: first flat workgroup index of this member (0 alone) */
    uint32_t m_tiles_x, m_tw, m_ho, m_pr, m_iw, m_iwd;
} SisrWgradDeepPlan;

typedef struct SisrWgradDesc {
    const float *x1, *x2, *pa, *pb, *pd, *ps, *pt;   /* conv-input operand                    */
    const float *g1, *g2, *qa, *qb, *qd, *qs, *qt;   /* output-gradient operand               */
    float *slab;            /* [n_slabs][slab_stride]: packed dW, layout [chunk][r][krow][co]       */
    float *bias_slab;       /* per-slab bias partial [CoutPad] at the same slab_stride, or NULL
                               (normally slab + slab_elems, slab_stride = slab_elems + CoutPad)    */
    int32_t N, H, W, Cin, Ho, Wo, Cout;
    int32_t KH, KW, stride, pad_y, pad_x;
    int32_t x_mode, pro_mode;
    const float *pro_slope_p;
    float pro_slope;
    int32_t g_mode, gpro_mode;
    const float *gpro_slope_p;
    float gpro_slope;
    /* plan (filled by sisr_wgrad_plan) */
    int32_t TH, TW, TN, tiles_y, tiles_x, n_groups, n_tiles;
    int32_t CK, PS, KROWP, n_chunk, CoutPad;
    int32_t NJ, NP, NT, TSTEP, TVALID;       /* co sub-tiles, pixel parts, row tiles          */
    int32_t grid_x, n_slabs, slab_elems, lds_bytes;
    uint32_t m_tiles_x, m_tiles_y, m_iw, m_twp, m_kw;   /* bf16 kernel: reciprocals as in SisrConvPlan */
    int32_t x_bf16, g_bf16;                  /* storage type of x1/x2 and of g1/g2 (0: fp32, 1: bf16) */
    int32_t mfma_split;                      /* fp32 operands, trunk geometry: as SisrConvDesc.mfma_split */
    int64_t slab_stride;                     /* set by the caller after planning              */
    SisrWgradDeepPlan deep;                  /* wgrad_deep.hip (enabled: sisr_conv2d_wgrad_bf16 runs the descriptor there)  */
} SisrWgradDesc;

int sisr_wgrad_plan(SisrWgradDesc *d, int32_t max_pixel_blocks);
/* bf16 matrix-core variants (v_mfma_f32_32x32x16_bf16, fp32 accumulate; activations fp32 or bf16 in HBM -- see the
 * *_bf16 storage flags -- converted while staged into LDS).  Same descriptors; `wpk` points at the bf16 image written by
 * sisr_weights_prepare (wbf_fwd / wbf_dgrad).  Requirements: Cin % 32 == 0, KH*KW <= 9, NHWC
 * operands; SISR_E_UNSUPPORTED otherwise (callers keep the fp32 kernels for those layers). */
int sisr_conv2d_plan_bf16(SisrConvDesc *d);
int sisr_conv2d_bf16(const SisrConvDesc *d, void *stream);
/* conv_deep.hip: fills d->deep for a descriptor whose geometry / modes are set (after sisr_conv2d_plan_bf16); returns
 * SISR_E_UNSUPPORTED (and leaves deep.enabled = 0) for geometries that family does not take: Cin % 32, Cout % 64, taps <= 3 x 3,
 * stride 1 | 2, NHWC bf16 in and out.  target_wg: workgroup slots a launch should fill through the K split (0: default 256).
 * sisr_conv2d_bf16 dispatches to it when deep.enabled && wdeep; sisr_conv2d_bf16_parts counts its partial rows. */
int sisr_conv2d_deep_plan(SisrConvDesc *d, int32_t target_wg, int32_t prefer_bn, int32_t classes);
/* prefer_bn: 0 = choose (64-cout tiles, two workgroups per CU, where there are plenty of pixel tiles; 128 otherwise), 64 | 128 = a
 * caller that knows its prologue is heavy (the two-tensor BatchNorm-backward forms cost ~500 vector instructions per chunk and
 * thread: a 128-cout tile has twice the MFMAs to hide them behind) asks for 128.  classes: 1, or 4 (see SisrConvDesc.wdeep_c). */
/* a fully filled descriptor (storage flags, modes, fusions, wdeep, deep_ws) will run on conv_deep.hip */
int sisr_conv2d_deep_eligible(const SisrConvDesc *d);
/* The generator's trunk geometry (3x3, 64 -> 64, stride 1, bf16 NHWC in and out, H % 8 == 0, W % 16 == 0; forward-type
 * prologue, no residual) runs on a persistent weights-in-registers kernel (conv_trunk.hip) behind sisr_conv2d_bf16;
 * sisr_conv2d_trunk_eligible tells whether a fully filled descriptor will.  That kernel writes ONE statistics partial
 * per workgroup instead of one per tile: sisr_conv2d_bf16_parts(d) = rows of stat_part / cnt_part (bnb_part) the launch
 * of `d` writes -- size those buffers with it (fill geometry, modes, storage flags, res / bnb pointers first). */
int sisr_conv2d_trunk_eligible(const SisrConvDesc *d);
int sisr_conv2d_bf16_parts(const SisrConvDesc *d);
/* The fp32 parity build has its own persistent kernel for that geometry (fp32 NHWC in and out, H % 8 == 0, W % 16 == 0;
 * conv_trunk_f32.hip, behind sisr_conv2d_f32): eligible returns 1 for the forward role, 2 for the data-gradient role;
 * sisr_conv2d_f32_parts(d) = rows of stat_part / cnt_part the launch writes (one per pair of workgroups). */
int sisr_conv2d_trunk_f32_eligible(const SisrConvDesc *d);
int sisr_conv2d_f32_parts(const SisrConvDesc *d);
/* rows of bnb_part (SisrConvDesc.bnb_*: fused BatchNorm-backward reductions) a data-gradient launch of `d` writes through
 * sisr_conv2d_f32 -- one per workgroup of the persistent fp32 trunk kernel; 0 when that kernel does not take `d` (the
 * generic fp32 kernel has no such epilogue: leave bnb_part NULL and run sisr_bn_bwd's own reduction). */
int sisr_conv2d_f32_bnb_parts(const SisrConvDesc *d);
/* bf16 build: the two 9x9 convolutions over a 3-channel NCHW fp32 image that produce 64 bf16 NHWC channels -- the
 * generator's first conv (model_generator.py:33) and the data gradient of its last one (model_generator.py:52, tanh'
 * as prologue) -- run on conv_thin.hip (bf16 MFMA operands like every other layer of that build) behind
 * sisr_conv2d_f32 when H % 16 == W % 16 == 0; tells whether a filled descriptor will. */
int sisr_conv2d_thin_eligible(const SisrConvDesc *d);
/* bf16 build: the generator's last conv (model_generator.py:52-53: 3x3, 64 -> 3, bf16 NHWC in, NCHW fp32 out, prologue
 * NONE / ACT, epilogue NONE / TANH) runs on conv_toimage.hip (1x1 GEMM onto 27 (cout, tap) columns + col2im gather)
 * behind sisr_conv2d_bf16; tells whether a filled descriptor will. */
int sisr_conv2d_toimage_eligible(const SisrConvDesc *d);
/* ... and with fp32 NHWC tensors (fp32 parity build; exact fp32 MFMA) behind sisr_conv2d_f32 */
int sisr_conv2d_toimage_f32_eligible(const SisrConvDesc *d);
int sisr_wgrad_plan_bf16(SisrWgradDesc *d, int32_t max_pixel_blocks);
/* wgrad_deep.hip: 3x3, pad 1, stride 1 | 2, Cin % 64 == 0, Cout % 64 == 0, NHWC bf16 operands (x prologue NONE / ACT / AFFINE_ACT,
 * gradient prologue any of the one- or two-tensor forms).  Call after sisr_wgrad_plan_bf16 (slab layout and sizes are shared with
 * the generic kernel); on success deep.enabled = 1 and sisr_wgrad_bf16_slabs answers deep.n_pb.  target_wg: 0 = default. */
int sisr_wgrad_deep_plan(SisrWgradDesc *d, int32_t target_wg);
/* a fully filled descriptor (operands, modes, storage flags) will run on wgrad_deep.hip */
int sisr_wgrad_deep_eligible(const SisrWgradDesc *d);
/* Several wgrad_deep.hip layers in one launch (a flat grid; deep.batch_first_wg, filled by the caller = the workgroups of the members
 * in front, n_cib * n_cob * n_pb each): table_host = n fully filled, eligible descriptors of one stride, all
 * with or all without a two-tensor gradient prologue; table_dev = the same bytes in device memory.  Plan each member with its share
 * of the chip as target_wg (sisr_wgrad_deep_plan): the batch then walks 3-4 times as many tiles per workgroup behind the same fixed
 * costs and writes a third of the slabs.  Results per layer are those of sisr_conv2d_wgrad_bf16 on the same plan. */
int sisr_wgrad_deep_batch(const SisrWgradDesc *table_host, const SisrWgradDesc *table_dev, int32_t n, void *stream);
/* The same idea for the persistent trunk kernel (wgrad_trunk.hip: 3x3, 64 -> 64, bf16 NHWC, H % 8 == 0, W % 16 == 0): n fully filled
 * descriptors that sisr_wgrad_trunk_eligible accepts, Cout = 64, ONE gradient-prologue kind (BNBWD or BNACT_BWD); wgs_per_layer
 * workgroups -- and slabs: rows of each descriptor's `slab` -- serve every layer, grid = n * wgs_per_layer (choose 256 / n: each
 * workgroup then walks its share of ONE layer's tiles back to back).  The kernel reads its own view of the descriptors:
 * sisr_wgrad_trunk_batch_args fills n * sisr_wgrad_trunk_batch_arg_bytes() bytes of HOST memory, the caller copies them to the device
 * and passes that copy as args_dev.  Per-tile arithmetic is that of sisr_conv2d_wgrad_bf16; only the partition into slabs differs. */
int sisr_wgrad_trunk_batch_arg_bytes(void);
int sisr_wgrad_trunk_batch_args(const SisrWgradDesc *descs, int32_t n, void *args_host);
int sisr_wgrad_trunk_batch(const SisrWgradDesc *descs, const void *args_dev, int32_t n, int32_t wgs_per_layer, void *stream);
/* ... and for the fp32-tensor trunk kernel (wgrad_trunk_f32.hip; exact fp32 or, with mfma_split set on every member, the split
 * contraction): descriptors sisr_wgrad_trunk_f32_eligible accepts, Cout = 64, one gradient-prologue kind, one mfma_split setting */
int sisr_wgrad_trunk_f32_batch_arg_bytes(void);
int sisr_wgrad_trunk_f32_batch_args(const SisrWgradDesc *descs, int32_t n, void *args_host);
int sisr_wgrad_trunk_f32_batch(const SisrWgradDesc *descs, const void *args_dev, int32_t n, int32_t wgs_per_layer, void *stream);
int sisr_conv2d_wgrad_bf16(const SisrWgradDesc *d, void *stream);
/* The trunk geometry with bf16 NHWC operands (x prologue NONE / ACT / AFFINE_ACT, gradient prologue BNBWD /
 * BNACT_BWD) runs on the persistent kernel of wgrad_trunk.hip behind sisr_conv2d_wgrad_bf16: one slab per workgroup,
 * whatever the plan's n_slabs says.  sisr_wgrad_bf16_slabs(d) = slabs the launch of a fully filled descriptor writes
 * -- size `slab` (rows of slab_stride floats) and the reduction with it. */
int sisr_wgrad_trunk_eligible(const SisrWgradDesc *d);
int sisr_wgrad_bf16_slabs(const SisrWgradDesc *d);
/* The same for the fp32 parity build: the trunk geometry with fp32 NHWC operands and H % 4 == 0, W % 16 == 0 runs on the
 * persistent exact-fp32 kernel of wgrad_trunk_f32.hip behind sisr_conv2d_wgrad_f32 (one slab per workgroup). */
int sisr_wgrad_trunk_f32_eligible(const SisrWgradDesc *d);
int sisr_wgrad_f32_slabs(const SisrWgradDesc *d);
/* bf16 build: the weight gradient of the generator's first conv (9x9 over the 3-channel NCHW fp32 image, bf16 NHWC
 * output gradient with no / activation-backward prologue, H % 8 == 0, W % 32 == 0) runs on wgrad_thin.hip (bf16 MFMA)
 * behind sisr_conv2d_wgrad_f32; tells whether a filled descriptor will (sisr_wgrad_f32_slabs accounts for it). */
int sisr_wgrad_thin_eligible(const SisrWgradDesc *d);
/* bf16 build: the weight gradient of the generator's last conv (3x3, 64 -> 3; bf16 NHWC x with prologue NONE / ACT, the
 * NCHW fp32 image gradient itself -- prologue NONE / TANH_BWD -- as g1 / g2, H % 8 == 0, W % 32 == 0, descriptor planned
 * by sisr_wgrad_plan_bf16 for the gradient padded to 4 channels) runs on wgrad_toimage.hip behind sisr_conv2d_wgrad_bf16:
 * no 4-channel NHWC copy of the gradient is needed then; same slab layout (sisr_wgrad_bf16_slabs accounts for it). */
int sisr_wgrad_toimage_eligible(const SisrWgradDesc *d);
/* ... and with fp32 NHWC activations (fp32 parity build; exact fp32 MFMA, descriptor planned by sisr_wgrad_plan) behind
 * sisr_conv2d_wgrad_f32; sisr_wgrad_f32_slabs accounts for it */
int sisr_wgrad_toimage_f32_eligible(const SisrWgradDesc *d);
int sisr_conv2d_wgrad_f32(const SisrWgradDesc *d, void *stream);
/* out[i] = sum_s slab[s][i], i < elems (also used for the bias slabs) */
int sisr_slab_reduce_f32(const float *slab, float *out, int32_t n_slabs, int64_t elems, int64_t lead_bf16, void *stream);
/* n_jobs independent reductions (arrays of n_jobs entries each, HOST memory: the jobs travel in the kernel arguments, eight per
 * launch) -- the weight gradients of one backward pass are summed by one launch instead of one per layer.  Same arithmetic and
 * order per job as sisr_slab_reduce_f32: bit-identical results. */
int sisr_slab_reduce_multi(const void *const *slabs, void *const *outs, const int32_t *n_slabs, const int64_t *elems,
                           const int64_t *lead_bf16, int32_t n_jobs, void *stream);
/* lead_bf16: the first lead_bf16 elements of every row are stored as bf16 at the row's start (what the persistent bf16
 * weight-gradient kernel writes: sisr_wgrad_bf16_slab_lead(d)), the rest -- the bias partials -- as fp32 at their float offset; 0: fp32 rows */
int64_t sisr_wgrad_bf16_slab_lead(const SisrWgradDesc *d);

/* ---- weights: spectral norm power iteration + packing (legacy torch.nn.utils.spectral_norm
 *      hook, model_generator.py:3; model_discriminator.py:2), multi-tensor: one launch serves
 *      every convolution of a network.  The descriptor TABLE lives in device memory. ---------- */
typedef struct SisrWeightDesc {
    const float *w_orig;      /* OIHW [Cout][Cin][KH][KW]                                     */
    float *u, *v;             /* spectral-norm buffers (updated in place when training) or NULL */
    float *u_used, *v_used;   /* copies of the u/v that define sigma (for backward) or NULL   */
    float *sigma;             /* [2] out: sigma (1.0 when u == NULL), 1 / sigma                */
    float *sn_work;           /* power-iteration scratch, >= ceil(Cout/16)*Cin*KH*KW + Cout floats (u != NULL) */
    float *wpk_fwd;           /* packed W/sigma for the forward conv, or NULL                 */
    float *wpk_dgrad;         /* packed flipped/transposed W/sigma for the data gradient, or NULL */
    int32_t Cout, Cin, KH, KW;
    int32_t training;         /* run the power iteration                                      */
    int32_t shuffle2;         /* conv feeds PixelShuffle(2): pack couts in (i,j)-major order  */
    /* forward packing */
    int32_t f_CK, f_PS, f_KROWP, f_n_chunk, f_CoutPad;
    /* data-gradient packing (roles of Cin/Cout swapped) */
    int32_t d_CK, d_PS, d_KROWP, d_n_chunk, d_CoutPad;
    /* stride-2 convs: the data gradient splits by output parity class c = (py, px) into four
     * stride-1 convs over dy with KH'xKW' taps; tap (r', s') of class c uses the forward tap
     * (c_R0y - 2 r', c_R0x - 2 s').  Packed like wpk_dgrad, one buffer per class (or NULL). */
    float *wpk_dcls[4];
    int32_t c_KH[4], c_KW[4], c_R0y[4], c_R0x[4];
    int32_t c_CK[4], c_PS[4], c_KROWP[4], c_n_chunk[4], c_CoutPad[4];
    /* bf16 images for the bf16-MFMA kernels: [chunk of CK in-channels][cout][tap*CK + ci] (or NULL) */
    void *wbf_fwd, *wbf_dgrad;
    int32_t bf_f_CoutPad, bf_d_CoutPad;
    int32_t bf_f_CK, bf_d_CK;   /* in-channel chunk of the bf16 images (32) */
    void *wbf_dcls[4];          /* bf16 images of the stride-2 parity classes (chunks of 32), or NULL: then wpk_dcls */
    int32_t bf_c_CoutPad[4];
    /* trunk geometry (Cin = 64, 3x3, CoutPad % 64 == 0): also write, right behind wbf_fwd / wbf_dgrad (same size again), the image
     * in the order the persistent trunk kernels load it -- [32-cout block][tap][k slice j: 16 in-channels][lane = kk * 32 + cout]
     * x 8 bf16 (in-channels (j >> 1) * 32 + (j & 1) * 16 + 8 kk ..) -- so that a wave's 16-byte loads are contiguous (1 KB per
     * instruction instead of 64 pieces 576 bytes apart: 1.1 us of every launch) */
    int32_t bf_f_lanes, bf_d_lanes;
    /* fp32 images, trunk geometry (Cin = Cout = 64, 3x3): also write, right behind wpk_fwd / wpk_dgrad (SISR_WLDS_WORDS more
     * 32-bit words), the weights in the persistent fp32-tensor conv kernel's LDS order -- [cout half][chunk][tap][cout 32][32 + 4
     * words] -- so that its weight fill is a plain 16-byte copy.  1: fp32 values; 2: split build -- words 0..15 of a row the RNE
     * bf16 heads of in-channels (2m, 2m + 1), words 16..31 the bf16 of what the heads leave; 0: none */
    int32_t f_ldsimg, d_ldsimg;
    /* conv_deep.hip images (or NULL): [32-channel chunk][tap row][cout][KW * 32 + 8] bf16, element kx * 32 + ci; the data-gradient
     * image has the roles of the channels swapped and the taps flipped; wdp_dcls = the four output-parity classes of a stride-2
     * convolution's data gradient (taps c_KH x c_KW, tap (r', s') = forward tap (c_R0y - 2 r', c_R0x - 2 s')).  wdp_scaled = 0 packs
     * W_orig itself (the kernels then apply 1 / sigma in their epilogue: SisrConvDesc.epi_scale_p), 1 packs W_orig / sigma */
    void *wdp_fwd, *wdp_dgrad;
    void *wdp_dcls[4];
    int32_t wdp_scaled;
    int32_t wdp_cls_kw;       /* 0: a class image has c_KW taps per row; 2: every class image is written in the KW = 2 row format */
} SisrWeightDesc;
#define SISR_WLDS_WORDS (2 * 2 * 9 * 32 * 36)

/* max_rows / max_cols: largest Cout and Cin*KH*KW over the table (the launch grids are sized from them) */
int sisr_weights_prepare(const SisrWeightDesc *table_dev, int32_t n, int32_t max_rows, int32_t max_cols,
                         void *stream);

/* sisr_weights_prepare = sisr_weights_sn (power iteration: u, v, sigma; sigma[1] = 1 / sigma) + sisr_weights_pack (every image but
 * the wdp_* ones).  sisr_weights_pack_deep writes the conv_deep.hip images (wdp_fwd / wdp_dgrad / wdp_dcls, 3x3 weights) from one
 * coalesced read of each 32 x 32-channel tile; max_cout / max_cin: largest channel counts over the table.  A caller that keeps
 * un-normalised images (wdp_scaled = 0) across the forwards between two optimizer steps runs sisr_weights_sn alone. */
int sisr_weights_sn(const SisrWeightDesc *table_dev, int32_t n, int32_t max_rows, int32_t max_cols, void *stream);
int sisr_weights_pack(const SisrWeightDesc *table_dev, int32_t n, int32_t max_rows, int32_t max_cols, void *stream);
int sisr_weights_pack_deep(const SisrWeightDesc *table_dev, int32_t n, int32_t max_cout, int32_t max_cin, void *stream);

/* Weight-gradient epilogue: packed dW (sum of slabs) -> OIHW gradient of w_orig, through the
 * spectral-norm quotient: dW_orig = (G - <G, W> u v^T) / sigma  (autograd of W = W_orig/sigma with
 * sigma = u.(W_mat v), u and v constant; spectral_norm.py compute_weight). */
typedef struct SisrWeightGradDesc {
    const float *dwpk;        /* [chunk][r][krow][CoutPad] reduced packed gradient            */
    const float *w_orig, *u_used, *v_used, *sigma;   /* u_used == NULL: no spectral norm      */
    float *grad;              /* OIHW out                                                     */
    const float *dbias_pk;    /* [CoutPad] reduced packed bias gradient or NULL               */
    float *grad_bias;         /* [Cout] out (original channel order) or NULL                  */
    int32_t Cout, Cin, KH, KW, shuffle2;
    int32_t CK, PS, KROWP, n_chunk, CoutPad;
    int32_t layout;           /* 0: fp32 wgrad slabs [chunk][r][krow][co]; 1: bf16 wgrad slabs
                                 [chunk32][tap][ci][co]                                          */
} SisrWeightGradDesc;

/* parts = the largest tile count of the table, tiles of a weight = ceil(Cout/32) * ceil(Cin/C) with C = 32 for
 * layout 1 (bf16 slabs) and CK otherwise; dot_work: >= parts*n floats of scratch */
/* workgroups along grid.y (= dot_work entries) one weight needs; `parts` >= the maximum over the table (host only) */
int sisr_weights_grad_tiles(const SisrWeightGradDesc *w);
int sisr_weights_grad(const SisrWeightGradDesc *table_dev, int32_t n, float *dot_work, int32_t parts, void *stream);
/* the same for a table of 3x3 weights with Cin % 32 == 0, layout 1 and no PixelShuffle permutation: whole 32 x 32 x 9 tiles, one
 * read of the packed gradient, the spectral-norm rank-one term as an elementwise pass afterwards (the discriminator's 8 convs:
 * 166 -> ~25 us per backward).  dot_work: n * ceil(max_cout / 32) * (max_cin / 32) floats. */
int sisr_weights_grad_fast(const SisrWeightGradDesc *table_dev, int32_t n, float *dot_work, int32_t max_cout, int32_t max_cin,
                           void *stream);

/* ---- BatchNorm2d (training) pieces that are not fused into the convolutions ------------------
 * finalize: merge the per-tile (mean, M2) partials (Chan et al.), produce the fused apply
 * constants scale = gamma*invstd, shift = beta - mean*scale, update the running statistics
 * (momentum 0.1, unbiased variance) -- nn.BatchNorm2d, model_generator.py:11,14,40. */
int sisr_bn_finalize(const float *stat_part, const float *cnt_part, int32_t n_tiles, int32_t C,
                     const float *gamma, const float *beta, float *running_mean, float *running_var,
                     float momentum, float eps, float *scale, float *shift, float *save_mean,
                     float *save_invstd, void *stream);
/* eval mode: scale/shift from the running statistics */
int sisr_bn_eval_consts(const float *gamma, const float *beta, const float *running_mean,
                        const float *running_var, float eps, int32_t C, float *scale, float *shift,
                        void *stream);

/* backward reductions over N*H*W for one BatchNorm (+ the activation that follows it):
 *   g  = (act_mode ? (z>0 ? dy : slope*dy) : dy),  z = scale[c]*x + shift[c]
 *   sum_g[c] = sum g ; sum_gx[c] = sum g*xhat,  xhat = (x-mean[c])*invstd[c]
 *   sum_slope = sum over z<=0 of dy*z   (gradient of a shared PReLU slope)
 * then the constants of SISR_PRO_BNBWD / SISR_PRO_BNACT_BWD:
 *   qa = gamma*invstd ; qb = -gamma*invstd^2*mean(g*xhat) ; qd = -qa*mean(g) - qb*mean
 * and dgamma = sum_gx, dbeta = sum_g, dslope (if act). */
typedef struct SisrBnBwdDesc {
    const float *dy, *x;                   /* NHWC [P][C]                                    */
    const float *scale, *shift, *mean, *invstd, *gamma;
    float *work;                           /* [grid][2*C+1] partials                          */
    float *qa, *qb, *qd;                   /* out [C]                                         */
    float *dgamma, *dbeta, *dslope;        /* out [C],[C],[1] (dslope may be NULL)            */
    int64_t P; int32_t C;
    int32_t act_mode;
    const float *slope_p; float slope;     /* device scalar slope, or NULL: use `slope`       */
    int32_t grid;                          /* filled by sisr_bn_bwd_plan                      */
    int32_t dy_bf16, x_bf16;               /* storage type of dy and x (0: fp32, 1: bf16)     */
} SisrBnBwdDesc;
int sisr_bn_bwd_plan(SisrBnBwdDesc *d);
int sisr_bn_bwd(const SisrBnBwdDesc *d, void *stream);
/* second half only: `work` holds d->grid partial rows [2*C+1] written by a conv epilogue (SisrConvDesc.bnb_part) */
int sisr_bn_bwd_finalize(const SisrBnBwdDesc *d, void *stream);
/* the same launch also carries a slab reduction (sisr_slab_reduce_f32's arguments) in additional workgroups: the
 * finishing step of a BatchNorm backward (a few latency-bound workgroups) and the slab sum of the weight gradient
 * computed just before it (hundreds of bandwidth-bound ones) are independent and adjacent in the backward schedule of
 * a residual block (model_generator.py:16-19 differentiated), so one launch replaces two -- 33 fewer per step.  Both
 * results are bit-identical to the separate launches. */
int sisr_bn_bwd_finalize_slab(const SisrBnBwdDesc *d, const float *slab, float *out, int32_t n_slabs, int64_t elems,
                              int64_t lead_bf16, void *stream);

/* elementwise: y = f(x1) + (pa ? pa[c]*x2 + pd[c] : x2)   over NHWC [P][C];
 * f = lrelu(., slope1_p ? *slope1_p : slope1) -- the residual add of BasicBlock.forward (model_generator.py:19) and the
 * long skip (model_generator.py:93) with the BatchNorm apply fused.  x2 may be NULL (y = f(x1)). */
/* Elementwise entry points take a storage word `dt`: bit k set = the k-th tensor argument (in argument order,
 * inputs then output) is bf16 instead of fp32 (SISR_DT(a, b, c) below); arithmetic is fp32 either way. */
#define SISR_DT(a, b, c) ((a) | ((b) << 1) | ((c) << 2))
int sisr_eltwise_res_affine(const float *x1, const float *slope1_p, float slope1, const float *x2,
                            const float *pa, const float *pd, float *y, int64_t P, int32_t C, int32_t dt /* x1, x2, y */,
                            void *stream);
/* sum over all elements where pre<=0 of dy*pre  -> out[0]  (gradient of a PReLU slope that is
 * not followed by... BatchNorm-free sites: model_generator.py:34,48) ; work: [grid] floats */
int sisr_prelu_slope_grad(const float *dy, const float *pre, int64_t n, float *work, float *out,
                          int32_t dt /* dy, pre */, void *stream);
/* y = a + b (same shape) */
int sisr_add(const float *a, const float *b, float *y, int64_t n, int32_t dt /* a, b, y */, void *stream);

/* ---- layout materialisation ------------------------------------------------------------------
 * NHWC [N][H][W][C] -> NCHW destination with row stride `dst_stride` floats per image (so a feature
 * map can be written straight into its slot of a concatenated [B, sum] feature vector), applying
 * y = lrelu(pa ? pa[c]*x + pd[c] : x, slope).  Used for D's flatten (model_discriminator.py:59),
 * MaskedVGG's taps (model_content_extractor.py:57-60) and forward_no_end (model_generator.py:86). */
int sisr_nhwc_to_nchw(const float *x, const float *pa, const float *pd, const float *slope_p,
                      float slope, float *y, int64_t dst_stride, int32_t N, int32_t H, int32_t W,
                      int32_t C, int32_t x_bf16, void *stream);
/* inverse gather: NCHW source (row stride src_stride per image) -> NHWC, y = x * (mask_pre ?
 * (lrelu'(mask_pre)) : 1) is NOT applied here: plain layout change. */
int sisr_nchw_to_nhwc(const float *x, int64_t src_stride, float *y, int32_t N, int32_t H, int32_t W,
                      int32_t C, int32_t y_bf16, void *stream);
/* NCHW gradient (C <= Cpad channels) -> NHWC [N][H][W][Cpad] with zero padding channels; out != NULL fuses the
 * tanh backward g = dy*(1 - out^2) of the generator's last layer (model_generator.py:54, nn.Tanh after the
 * final conv).  Feeds the bf16 weight gradient of that 3-channel conv. */
int sisr_nchw_grad_to_nhwc4(const float *dy, const float *out, float *g, int32_t N, int32_t C, int32_t H,
                            int32_t W, int32_t Cpad, void *stream);

/* ---- MaxPool2d(2,2) of the VGG19 stack (model_content_extractor.py:43; floors odd sizes), NHWC.
 * Pooling commutes with the (monotonic) ReLU in front of it, so the forward pools the RAW conv
 * output and the consumer applies ReLU lazily; the backward fuses MaxPool' and ReLU':
 *   dx[argmax of the 2x2 window] = (x[argmax] > 0) ? dy : 0, all other positions 0
 * (first maximum in row-major window order, like ATen). */
int sisr_maxpool2_fwd(const float *x, float *y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dt /* x, y */,
                      void *stream);
int sisr_maxpool2_relu_bwd(const float *dy, const float *x, float *dx, int32_t N, int32_t H, int32_t W,
                           int32_t C, int32_t dt /* dy, x, dx */, void *stream);
/* out = a + (ref > 0 ? b : 0)   (a may be NULL: out = masked b) -- merges a tap gradient into the
 * gradient of the ReLU it was taken behind (MaskedVGG's in-place-ReLU aliasing). */
int sisr_add_relu_masked(const float *a, const float *b, const float *ref, float *out, int64_t n,
                         int32_t dt /* a, b, ref, out */, void *stream);

/* ---- fully connected layers of D (nn.Linear, model_discriminator.py:47-53); weight-streaming,
 *      HBM-bound on W[Nout][K] (75-302 MB).  x operand: lrelu(x, in_slope) applied on load. ------ */
/* y[b][n] = act( sum_k lrelu(x[b][k]) * W[n][k] + bias[n] ),  epi: 0 none, 1 sigmoid; B <= 16 */
int sisr_fc_forward(const float *x, float in_slope, const float *W, const float *bias, float *y,
                    int32_t B, int32_t K, int32_t Nout, int32_t epi, void *stream);
/* dx[b][k] = sum_n dy[b][n] * W[n][k]; work: [splits][B][K] floats, splits = sisr_fc_dgrad_splits */
int sisr_fc_dgrad_splits(int32_t K, int32_t Nout);
int sisr_fc_dgrad(const float *dy, const float *W, float *dx, float *work, int32_t B, int32_t K,
                  int32_t Nout, void *stream);
/* dW[n][k] = sum_b dy[b][n] * lrelu(x[b][k], in_slope); db[n] = sum_b dy[b][n] */
int sisr_fc_wgrad(const float *dy, const float *x, float in_slope, float *dW, float *db, int32_t B,
                  int32_t K, int32_t Nout, void *stream);
/* The whole classifier head of D (model_discriminator.py:47-53: Linear(K, N) -> LeakyReLU -> Linear(N, 1) -> Sigmoid) on the
 * exact-fp32 matrix instruction (fc_head.hip), B <= 16, K % 16 == 0 (forward) / K % 64 == 0 (data gradient), N % 128 == 0:
 *   forward:  h1 [B][N] = x W1^T + b1 (pre-activation, kept for the backward), y [B] = sigmoid(lrelu(h1, slope) . W2 + b2);
 *             ws: sisr_fc_head_ws_floats(N) floats.  W2 == NULL: first layer only.
 *   backward: from g [B] = dL/dy: d1 [B][N] = dL/dh1, dW2 [N], db2 [1], db1 [N]  (everything that is not W1-sized);
 *   sisr_fc1_dgrad: dx [B][K] = d1 W1.   dW1 = d1^T x stays with sisr_fc_wgrad (an outer product: write-bound). */
int sisr_fc_head_ws_floats(int32_t N);
int sisr_fc_head_forward(const float *x, const float *W1, const float *b1, const float *W2, const float *b2, float slope,
                         float *h1, float *y, float *ws, int32_t B, int32_t K, int32_t N, void *stream);
int sisr_fc_head_backward(const float *g, const float *y, const float *h1, const float *W2, float slope, float *d1,
                          float *dW2, float *db2, float *db1, int32_t B, int32_t N, void *stream);
int sisr_fc1_dgrad(const float *d1, const float *W1, float *dx, int32_t B, int32_t K, int32_t N, void *stream);
/* dW[n][k] = scale * sum_b dy[b][n] * x[b][k] for MANY rows (B <= 256; K % 128 == 0, N % 64 == 0) on the exact-fp32 matrix instruction:
 * the data-parallel form of the classifier head's weight gradient -- ranks exchange the two rank-16 FACTORS (dy, x: N_ranks x 1.2-4.8 MB
 * through an all-gather) instead of all-reducing their 75-302 MB product, and every rank forms the mean itself (scale = 1 / ranks)
 * from all N_ranks x 16 rows: the same bits on every rank. */
int sisr_fc_wgrad_rows(const float *dy, const float *x, float scale, float *dW, int32_t B, int32_t K, int32_t N, void *stream);
/* elementwise helpers for the tiny FC activations: dy_pre = dy * act'(...) */
int sisr_act_bwd(const float *dy, const float *ref, float *out, int64_t n, int32_t kind, float slope,
                 void *stream);   /* kind 0: leaky (ref = pre-activation), 1: sigmoid (ref = output) */

/* ---- dataset transform ---------------------------------------------------------------------------
 * replaces, for a batch of decoded images, the per-image transform of config.py:225-231
 *   transforms.Compose([transforms.Resize(image_size_hr[1:]), transforms.ToTensor(), transforms.Normalize(.5, .5)])
 * i.e. PIL.Image.resize(BILINEAR) (Pillow's ImagingResample, 8 bits per channel: anti-aliased separable triangle filter,
 * 22-bit fixed-point coefficients, rounding to uint8 after each pass), uint8 -> float32 / 255, (t - mean) / std.
 * Integer arithmetic throughout the resize: bit-exact with Pillow.
 * sisr_resize_coeffs: HOST function; returns the taps per output index (ksize) and, with non-NULL tables, fills
 *   bounds[out_size][2] = (first input index, tap count) and kk[out_size][ksize] for one axis.  ksize > 64 (a down-scale
 *   above ~31x) is SISR_E_UNSUPPORTED in BOTH call modes, so the size query already refuses what the fill would.
 * sisr_resize_u8_normalize: src [N][H0][W0][C] uint8 -> dst [N][C][H][W] float32 with device copies of the two tables. */
int sisr_resize_coeffs(int32_t in_size, int32_t out_size, int32_t *bounds, int32_t *kk);
int sisr_resize_u8_normalize(const unsigned char *src, float *dst, int32_t N, int32_t H0, int32_t W0, int32_t C,
                             int32_t H, int32_t W, const int32_t *bx, const int32_t *kx, int32_t ksx,
                             const int32_t *by, const int32_t *ky, int32_t ksy, float mean, float stdv, void *stream);

/* ---- bicubic degradation, align_corners=True, A=-0.75, clamp to [-1,1]
 *      (utils.py:16-31: F.interpolate(..., 'bicubic', align_corners=True) + _crop_lr) -------- */
int sisr_bicubic_fwd(const float *x, float *y, int32_t NC, int32_t H, int32_t W, int32_t Ho,
                     int32_t Wo, int32_t clamp, void *stream);
/* dx = transpose of the interpolation applied to dy (masked by the clamp when y_clamped is given); gather form:
 * every input pixel sums the output gradients that read it in a fixed order -- deterministic, no atomics */
int sisr_bicubic_bwd(const float *dy, const float *y_clamped, float *dx, int32_t NC, int32_t H,
                     int32_t W, int32_t Ho, int32_t Wo, void *stream);

/* ---- misc ---------------------------------------------------------------------------------- */
/* ---- optimizer: fused multi-tensor Adam --------------------------------------------------
 * One launch performs torch.optim.Adam's update (amsgrad=False, maximize=False; config.py:292-294, stepped at
 * train.py:75,108) for every parameter of a network.  Table in DEVICE memory; block_start = running sum of
 * sisr_adam_blocks(numel) over the preceding entries; total_blocks = that sum over all entries.  The caller passes
 * lr (after any LambdaLR factor, config.py:170-180) and the bias corrections 1 - beta^t of the step being taken. */
typedef struct SisrAdamDesc {
    float *p, *m, *v;          /* parameter, exp_avg, exp_avg_sq (updated in place)            */
    const float *g;            /* gradient                                                      */
    int64_t numel;
    int64_t block_start;
} SisrAdamDesc;
int64_t sisr_adam_blocks(int64_t numel);
int sisr_adam_step(const SisrAdamDesc *table_dev, int32_t n, int64_t total_blocks, double lr, double beta1, double beta2,
                   double eps, double weight_decay, double bias_corr1, double bias_corr2, void *stream);

/* sizeof() of the descriptor structs in declaration order (Conv, Wgrad, Weight, WeightGrad,
 * BnBwd, ConvPlan, DeepPlan, WgradDeepPlan) so a binding can verify its mirror of this header; returns the count. */
int sisr_struct_sizes(int32_t *out, int32_t cap);
int sisr_device_info(int32_t *n_cu, int32_t *lds_per_cu, char *arch, int32_t arch_len);
int sisr_mfma_selftest(float *out_dev /* >= 32*32 floats */, void *stream);
int sisr_tr16_selftest(int16_t *out_dev /* >= 64*8 shorts */, void *stream);
/* fetch-and-clear the calling thread's pending HIP error (e.g. after an abandoned stream capture) so that it is
 * not attributed to the next launch; returns the code that was pending (0: none) */
int sisr_clear_last_error(void);
const char *sisr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SISR_HIP_H */
