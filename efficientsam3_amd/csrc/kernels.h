// Host-callable launchers of the non-GEMM kernels (definitions in *.hip).
// dtype: 0 = f32 activations, 1 = bf16 activations.  All tensors NHWC / row-major.
#pragma once
#include <vector>

#include "esam3_common.h"

int esam3_gemm_pad_n(int N);
int esam3_gemm_pad_k(int K, int elem_size);
// name of the kernel chosen by the most recent esam3_launch_gemm on this thread (then cleared)
const char* esam3_take_last_gemm_kernel();
void esam3_note_gemm_kernel(const char* name);
// allow `bytes` of dynamic LDS for `kernel` on the current device (once per device and kernel)
int esam3_allow_dyn_lds(const void* kernel, int bytes);
// decoder_fused.hip: "image attends to the tokens" of the two-way transformer in one kernel (q_proj + attention over T <= 16
// prompt tokens + out_proj + residual + LayerNorm), bf16, 8 heads x 16
// out[r] = x[r] W^T + bias + table[r mod P] for 256 -> 256 channels, bf16, weights resident in LDS (decoder_fused.hip)
bool esam3_rowlin256_ok(int dtype, int64_t rows, int N, int K, int P);
int esam3_launch_rowlin256(const void* x, const void* w, int kp, const float* bias, const void* table, int P, void* out, int64_t rows,
                           hipStream_t s);
// token side of the two-way transformer + the output heads as three per-prompt kernels (decoder_fused.hip); a Linear is
// {packed bf16 weight [N][ldw], fp32 bias or null, ldw}
struct esam3_tok_lin { const void* w; const float* bias; int ldw; };
bool esam3_tok_fused_ok(int dtype, int T);
// lin: self_attn q, k, v, out_proj, cross_attn_token_to_image.q_proj
int esam3_launch_tok_a(float* q32, const float* t32, void* tq, const esam3_tok_lin lin[5], const float* g1, const float* b1, float eps,
                       int Bp, int T, int first, hipStream_t s);
// lin: cross_attn_token_to_image.out_proj, mlp.lin1, mlp.lin2, cross_attn_image_to_token.k_proj, .v_proj, final_attn q_proj (or unused)
// scratch: esam3_tok_b_scratch_bytes(Bp) (two launches: the MLP spread over 8 workgroups per prompt, then the tail)
int64_t esam3_tok_b_scratch_bytes(int Bp);
int esam3_launch_tok_b(float* q32, const float* t32, const void* ta, void* tk, void* tv, void* tq_final, const esam3_tok_lin lin[6],
                       const float* g2, const float* b2, const float* g3, const float* b3, float eps, void* scratch, int Bp, int T,
                       hipStream_t s);
// xo: final_attn out_proj; mlp[r * 3 + layer]: r = 0..3 hypernetwork MLPs, 4 IoU head, 5 object-score head
int esam3_launch_tok_d(float* q32, const void* ta, void* hs, const esam3_tok_lin& xo, const float* gf, const float* bf, float eps,
                       const esam3_tok_lin mlp[18], void* hyper, float* iou, void* obj, int Bp, int T, hipStream_t s);
// token -> image attention (<= 16 queries, 8 heads x 16) on the matrix cores; k / v may be the two halves of one [rows][256] tensor
bool esam3_t2i_mfma_ok(int dtype, int Nq, int Nk, int heads, int hd);
int64_t esam3_t2i_mfma_scratch_floats(int Bp, int Nq, int Nk);
int esam3_launch_t2i_mfma(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int Bp, int Nq, int Nk,
                          float* scratch, hipStream_t s);
int esam3_launch_attn_t2i_merge(int dtype, const float* parts, void* o, int B, int Nq, int nparts, hipStream_t s);
bool esam3_i2t_fused_ok(int dtype, int P, int T, int heads, int hd, int C);
int esam3_launch_i2t_fused(const void* x, void* out, const void* wq, int kpq, const float* bq, const void* peq, const void* wo, int kpo,
                           const float* bo, const float* gamma, const float* beta, float eps, const void* tk, int ldk, const void* tv,
                           int ldv, int Bp, int P, int T, hipStream_t s);
int esam3_launch_gemm(int dtype, const GemmParams& p, hipStream_t stream);
// K ordering of packed dense-conv weights: korder (see GemmParams) and the packed k index of (tap, c)
int esam3_conv_korder(int cin, int ksize, int elem_size);
int esam3_conv_k_index(int cin, int ksize, int elem_size, int tap, int c);
// 3x3 convs with 32 / 64 output channels on a zero-bordered input (conv3x3_narrow.hip): eligibility, the element
// index of weight (n, tap, c) in the fragment-ordered bf16 weight array the kernel expects in p.Wt, the launcher
bool esam3_conv3x3_narrow_ok(int dtype, int N, int Cin, int H, int W, int in_pad, int out_pad, int stride, bool has_res);
int64_t esam3_conv3x3_narrow_windex(int N, int n, int tap, int c);
int esam3_launch_conv3x3_narrow(const GemmParams& p, hipStream_t stream);
// up-conv (ConvT k2s2 composed with the 3x3 + 1x1 that follow) with 32 output channels per parity class, same file:
// eligibility, weight index of (n, class, tap kh*2+kw, c) in the staged order, launcher (p.convt_cout = 32, p.H / p.W = input size)
// Up-conv composition (engine.hip): ConvT k2s2 (+ its 1x1) followed by a 3x3 conv as four 2x2 convs on the ConvT's input, on host arrays;
// and the K position of (tap = kh*2 + kw, input channel ci) in a packed row of gemm256p's up-conv gather (64-channel chunk major).
void esam3_compose_upconv_host(const float* wt, const float* bt, const float* w3, const float* b3, int cin, int cm, int co,
                               std::vector<float>& w, std::vector<float>& bias, std::vector<float>& corr);
inline int64_t esam3_upconv_kindex(int tap, int ci) { return (int64_t)(ci / 64) * 256 + tap * 64 + ci % 64; }
bool esam3_upconv_narrow_ok(int dtype, int Cout, int Cin, int H, int W);
int64_t esam3_upconv_narrow_windex(int n, int cls, int tap, int c);
int esam3_launch_upconv_narrow(const GemmParams& p, hipStream_t stream);

// Fused pointwise MLP out = res + W2 act(W1 x + b1) + b2 (fused_mlp.hip, bf16): eligibility, the hidden index at position
// `pos` of a 32-block of the second layer's packed K order, launcher (w1 [Hid][Cin], w2perm [Cout][Hid] in that order)
bool esam3_fused_mlp_ok(int dtype, int Cin, int Hid, int Cout);
int esam3_fused_mlp_kperm(int pos);
int esam3_launch_fused_mlp(const void* x, int ldx, const void* w1, const float* b1, const void* w2perm, const float* b2, const void* res,
                           int ldr, void* out, int ldo, int64_t M, int Cin, int Hid, int Cout, int act, hipStream_t s);

// E0: stem 3x3/s2 conv on the NCHW fp32 network input -> NHWC T, + bias + Hardswish.
int esam3_launch_stem(int dtype, const float* img_nchw, const float* w /*[27][Cout]*/,
                      const float* bias, void* out, int B, int H, int W, int Cout, int act,
                      hipStream_t s);

// EfficientViT input stem fused: stem conv (w0 [27][16], b0) + Hardswish -> Residual(DSConv: depthwise 3x3 (wd [9][16], bd)
// + Hardswish -> pointwise 16 -> 16 (wp: activation dtype, row stride ldw, bp)).  16 channels only.
int esam3_launch_stem_dsconv(int dtype, const float* img_nchw, const float* w0, const float* b0, const float* wd,
                             const float* bd, const void* wp, int ldw, const float* bp, void* out, int B, int H, int W,
                             hipStream_t s, int variant = 0);

// depthwise k x k (k in {3,5}), stride in {1,2}, pad k/2.  w: fp32 [k*k][C]; bias fp32 or null.
int esam3_launch_dwconv(int dtype, const void* in, int ld_in, const float* w, const float* bias,
                        void* out, int ld_out, int B, int H, int W, int C, int ksize, int stride,
                        int act, hipStream_t s);

// grouped 1x1 conv with `gs` in/out channels per group (LiteMLA aggreg.0.1), no bias.
// w: fp32 [C/gs][gs out][gs in].
int esam3_launch_grouped_pw(int dtype, const void* in, int ld_in, const float* w, void* out,
                            int ld_out, int64_t rows, int C, int gs, hipStream_t s);

// LiteMLA ReLU linear attention (ops.py:584-621).  ms: [B][N][ld] with `groups` groups of
// (q|k|v) x dim channels; kv: fp32 scratch [B][groups][dim+1][dim]; out: [B][N][ld_out].
// kv: fp32 scratch of esam3_lite_mla_scratch_floats() elements
int64_t esam3_lite_mla_scratch_floats(int B, int N, int groups, int dim);
int esam3_launch_lite_mla(int dtype, const void* ms, int ld, void* out, int ld_out, float* kv,
                          int B, int N, int groups, int dim, hipStream_t s);

// bilinear resize, align_corners=False (F.interpolate), NHWC T -> NHWC T.
int esam3_launch_resize_bilinear(int dtype, const void* in, void* out, int B, int IH, int IW,
                                 int OH, int OW, int C, hipStream_t s);

// the same interpolation applied to the output of a neck level's first layer computed on the un-resized map: in
// [B][IH][IW][taps*C] (taps = 4: ConvT k2s2 tap-major blocks, pixel-shuffled to a 2x larger map; taps = 1: 1x1 conv) + bias
// + activation -> out [B][s*OH (+2)][s*OW (+2)][C], optionally inside a 1-pixel zero border
int esam3_resize_axis_tables_host(int in_size, int out_size, int* first, int* count, float* frac);  // host-only: the row kernel's axis maps
int esam3_launch_resize_shuffle(int dtype, const void* in, const float* bias, void* out, int B, int IH, int IW, int OH, int OW,
                                int C, int taps, int act, int out_pad, hipStream_t s);

// y = act(LN(x (+ res))) over the last dim C (biased variance), one wavefront per row.
// the same with separate row dtypes (0 f32, 1 bf16): fp32 rows in / bf16 rows out for an fp32 residual stream, and back
int esam3_launch_layernorm_io(int in_dtype, int out_dtype, const void* x, const void* res, const float* gamma, const float* beta,
                              void* out, int64_t rows, int C, float eps, int act, hipStream_t s);
int esam3_launch_layernorm(int dtype, const void* x, const void* res, const float* gamma,
                           const float* beta, void* out, int64_t rows, int C, float eps, int act,
                           hipStream_t s);

// out[bp][p][c] = in[src_img[bp]][p][c] + cbias[c] (+ dense[bp][p][c])
int esam3_launch_gather_add(int dtype, const void* in, const int* src_img, const float* cbias,
                            const void* dense, void* out, int Bp, int64_t P, int C, hipStream_t s);

// softmax attention, few queries vs many keys (token -> image):  q [Bq][Nq][ldq] etc.
// `scratch`: fp32 workspace of esam3_attn_scratch_floats(...) elements (0 = the shape has no tiled variant) or null;
// with it, <= 16 queries x 8 heads x 16 against >= 1024 keys run on the LDS-tiled kernel.
int64_t esam3_attn_scratch_floats(int B, int Nq, int Nk, int heads, int hd);
int esam3_launch_attn(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v,
                      int ldv, void* o, int ldo, int B, int Nq, int Nk, int heads, int hd,
                      float* scratch, hipStream_t s);
// softmax attention, many queries vs few keys (image -> token), Nk <= 64.
int esam3_launch_attn_fewkeys(int dtype, const void* q, int ldq, const void* k, int ldk,
                              const void* v, int ldv, void* o, int ldo, int B, int Nq, int Nk,
                              int heads, int hd, hipStream_t s);

// tokens[bp] = [obj | iou | mask x4 | sparse prompt embeddings]  (prompt_encoder.py:74-118,
// mask_decoder.py:177-197).  coords: [Bp][Np][2] in network pixels (NOT yet +0.5),
// labels [Bp][Np] in {-1,0,1,2,3}; a pad point (label -1) is appended when `pad`.
int esam3_launch_build_tokens(int dtype, const float* out_tokens /*[6][256]*/,
                              const float* coords, const int* labels, const float* gauss /*[2][128]*/,
                              const float* point_emb /*[4][256]*/, const float* not_a_point /*[256]*/,
                              void* tokens, int Bp, int Np, int pad, float img_size, hipStream_t s);

// masks[bp][k][p] = sum_c hyper[bp][k][c] * up[bp][p][c]   (k < 4, c < 32), fp32 out.
// MobileCLIP-S0 text encoder pieces (rows = B*S tokens)
int esam3_launch_text_embed(int dtype, const int64_t* tokens, const float* table, const float* pos, void* x,
                            float* embeds_sbd, int B, int S, int D, int vocab, hipStream_t s);
int esam3_launch_seq_dwconv(int dtype, const void* x, const float* w /*[KW][D]*/, const float* bias, void* out, int B,
                            int S, int D, int KW, hipStream_t s);
int esam3_launch_text_attn(int dtype, const void* qkv /*[B*S][3*heads*hd]*/, void* out, int B, int S, int heads, int hd,
                           int causal, hipStream_t s);
int esam3_launch_bsc_to_sbc_f32(int dtype, const void* x, float* out, int B, int S, int C, hipStream_t s);
// w = {conv0.w, conv0.b, ln1.w, ln1.b, conv3.w, conv3.b, ln4.w, ln4.b, conv6.w, conv6.b} (device fp32)
int esam3_launch_mask_embed(int dtype, const float* mask, const float* const* w, void* out, int Bp, int in_size,
                            int emb_size, hipStream_t s);
// bf16: ConvTranspose2d(64 -> 32, k2 s2) of u1 [Bp][S*S][64] (weights wt [>=128][kp], n = tap*32 + co) + bias + feat
// [B][4*S*S][32] gathered by img_of[bp], GELU, product with hyper [Bp][4][ld_h] -> masks [Bp][4][4*S*S] fp32
int esam3_launch_upscale_mask(const void* u1, const void* wt, int kp, const float* bias, const void* feat, const int* img_of,
                              const void* hyper, int ld_h, float* masks, int Bp, int S, hipStream_t s);
int esam3_launch_mask_product(int dtype, const void* hyper, int ld_h, const void* up, float* masks,
                              int Bp, int64_t P, int C, hipStream_t s);

// output selection (mask_decoder.py:142-163,244-292).  all_masks [Bp][4][P] fp32, all_iou [Bp][4]
// (T).  multimask: out = masks 1..3; else dynamic stability fallback.  counters: int[2*Bp] scratch.
int esam3_launch_select_masks(int dtype, const float* all_masks, const void* all_iou, int ld_iou,
                              float* out_masks, float* out_iou, int* counters, int Bp, int64_t P,
                              int multimask, float delta, float thresh, hipStream_t s);

// post-processing (sam1_utils.py:77-119): fill background holes (8-connected components of
// score <= thr with area <= max_area) with thr + 10.  labels: int32 scratch [n][H*W] x2.
int esam3_launch_fill_holes(const float* in, float* out, int* labels, int* areas, int n, int H,
                            int W, float thr, float max_area, hipStream_t s);

// bilinear upsample of fp32 masks [n][IH][IW] -> [n][OH][OW]; optionally threshold to u8.
int esam3_launch_upsample_masks(const float* in, float* out_f32, uint8_t* out_u8, int n, int IH,
                                int IW, int OH, int OW, float thr, hipStream_t s);

// out = a + b (elementwise, activation dtype)
int esam3_launch_add(int dtype, const void* a, const void* b, void* out, int64_t n, hipStream_t s);
// out (bf16) = a (f32) + b (f32, or null): the bf16 GEMM-side copy of an fp32 token stream (+ its positional tokens)
int esam3_launch_add_f32_to_bf16(const float* a, const float* b, void* out, int64_t n, hipStream_t s);
// out[i] = f32(in[i * ld]) for i < n
int esam3_launch_strided_to_f32(int dtype, const void* in, int ld, float* out, int64_t n, hipStream_t s);
// clamp fp32 buffer in place to [lo, hi]
int esam3_launch_clamp(float* x, int64_t n, float lo, float hi, hipStream_t s);
// dtype conversion helpers
int esam3_launch_cast_to_f32(int dtype, const void* in, float* out, int64_t n, hipStream_t s);
int esam3_launch_cast_from_f32(int dtype, const float* in, void* out, int64_t n, hipStream_t s);
// NHWC T -> NCHW fp32 (API boundary helper for callers that need the reference layout)
int esam3_launch_nhwc_to_nchw_f32(int dtype, const void* in, float* out, int B, int H, int W, int C,
                                  hipStream_t s);
// uint8 HWC -> fp32 NCHW, x/255 then (x-0.5)/0.5
int esam3_launch_preprocess_u8(const uint8_t* in, float* out, int B, int H, int W, hipStream_t s);
// ViT-H teacher: patch rows for the 14x14 patch-embedding GEMM, in-place axial RoPE on q|k, and
// softmax attention over ws x ws windows (ws = H: global) with head dim 64
int esam3_launch_patchify(int dtype, const float* img_nchw, void* a, int B, int S, int P, int ldk, hipStream_t s);
int esam3_launch_vit_rope(int dtype, void* qkv, const float* cos_sin /*[ws*ws][32][2]*/, int64_t rows, int H, int W, int ws,
                          int heads, hipStream_t s);
// rope: optional [ws*ws][32][2] (cos, sin) table applied to q and k (on the fly in the bf16 MFMA kernel, as an
// in-place pass over qkv before the fp32 kernel)
int esam3_launch_attn_window(int dtype, void* qkv, int ld, int q_off, int k_off, int v_off, void* out, int ldo, int B,
                             int H, int W, int ws, int heads, int hd, const float* rope, hipStream_t s);
// TinyViT window attention, head dim 32: qkv [B][H][W][heads*96] (q|k|v per head), pad_qkv [heads*96] (T),
// bias [heads][ws*ws] fp32 indexed by |dy|*ws+|dx|; out [B][H][W][heads*32]
int esam3_launch_window_attn(int dtype, const void* qkv, int ld, const void* pad_qkv, const float* bias, void* out,
                             int ldo, int B, int H, int W, int heads, int ws, hipStream_t s);
// in-place squeeze-excite on x [B][HW][C]; sums: esam3_squeeze_excite_scratch_floats() fp32, gate: [B][C] fp32; w1 [R][C], w2 [C][R] (device fp32)
int64_t esam3_squeeze_excite_scratch_floats(int B, int HW, int C);
int esam3_launch_squeeze_excite(int dtype, void* x, int ld, float* sums, float* gate, const float* w1,
                                const float* b1, const float* w2, const float* b2, int B, int HW, int C, int R,
                                hipStream_t s);
// B images of one size: in [B][H][W][3] u8 -> out [B][3][OH][OW] f32
int esam3_launch_resize_aa_u8(const uint8_t* in, int B, int H, int W, float* out, int OH, int OW, hipStream_t s, int pixel_bytes = 3);
// zero the 1-pixel border of a [B][Hp][Wp][C] tensor (Hp = H+2, Wp = W+2)
int esam3_launch_zero_border(int dtype, void* x, int B, int Hp, int Wp, int C, hipStream_t s);
// fused MBConv (mbconv_fused.hip): returns LDS bytes needed, 0 if the shape is unsupported
size_t esam3_mbconv_fused_lds(int dtype, int Cin, int Cmid, int Cout, int stride);
// true when esam3_launch_mbconv_fused runs this shape on the v2 kernel (bf16, EfficientViT-B1 stage 1-3 shapes)
bool esam3_mbconv_fused2_ok(int dtype, int Cin, int Cmid, int Cout, int stride);
// `residual`: bit 0 = identity shortcut; bit 1 (v2 shape 64 -> 64, stride 1 only) = TinyViT MBConv: GELU instead of Hardswish after
// the expand and the depthwise conv, and a GELU after the shortcut add (tiny_vit.py:73-108)
int esam3_launch_mbconv_fused(int dtype, const void* x, void* out, const void* w1, int Kp1, const float* b1,
                              const float* wd, const float* bd, const void* w2, int Kp2, const float* b2, int B,
                              int H, int W, int Cin, int Cmid, int Cout, int stride, int residual,
                              hipStream_t stream);

// round-4 fused MBConv (evit_fused.hip, bf16): depthwise phase on the matrix cores, channels up to 256; same arguments as
// esam3_launch_mbconv_fused (w1 / w2 packed [N][Kp] bf16, wd fp32 [9][Cmid])
bool esam3_mbconv3_ok(int dtype, int Cin, int Cmid, int Cout, int stride);
bool esam3_patch_merging_fused_ok(int dtype, int Cin, int Cout);
int esam3_launch_mbconv3(const void* x, void* out, const void* w1, int Kp1, const float* b1, const float* wd, const float* bd,
                         const void* w2, int Kp2, const float* b2, int B, int H, int W, int Cin, int Cmid, int Cout, int stride,
                         int residual, hipStream_t stream);

// fused LiteMLA context module (evit_fused.hip, bf16, dim 16, C = 128 / 256): out = x + proj(BN)(relu linear attention of the
// two-scale qkv); scratch sizes from esam3_mla_fused_scratch
bool esam3_mla_fused_ok(int dtype, int C, int dim);
void esam3_mla_fused_scratch(int B, int H, int W, int C, size_t* qms_bytes, size_t* kvp_bytes, size_t* tab_bytes);
int esam3_launch_mla_fused(const void* x, void* out, const void* wqkv, int Kpq, const float* wdw, const void* wgrp, int Kpg,
                           const void* wproj, int Kpp, const float* bproj, void* qms, float* kvp, void* tab, int B, int H, int W,
                           int C, hipStream_t stream);

// bf16 MFMA flash attention, heads x 32, no mask / bias (returns 1 when the shape is not eligible)
int esam3_launch_attn_mfma32(const void* q, int ldq, int q_off, const void* kv, int ldk, int k_off, int v_off, void* out,
                             int ldo, int B, int Nq, int Nk, int heads, hipStream_t s);
// the same for FEW queries against many keys (Nq <= ~1k): 32 queries per block, the four waves split the key
// tiles and merge (m, l, O) through LDS; optional key mask [B][Nk] and separable bias (layout of mha_core)
int esam3_launch_attn_mfma32_splitk(const void* q, int ldq, int q_off, const void* kv, int ldk, int k_off, int v_off,
                                    void* out, int ldo, int B, int Nq, int Nk, int heads, const uint8_t* key_mask,
                                    const float* bias_y, const float* bias_x, int Hk, int Wk, int bias_q0, hipStream_t s);
// ---- PCS text-grounding detector (kernels_pcs.hip) ---------------------------------------------------------
int esam3_launch_mha_core(int dtype, const void* q, int ldq, int q_off, const void* kv, int ldk, int k_off, int v_off,
                          void* out, int ldo, int B, int Nq, int Nk, int heads, const uint8_t* key_mask,
                          const float* bias_y, const float* bias_x, int Hk, int Wk, int bias_q0, hipStream_t s);
int esam3_launch_pcs_prompt(int dtype, const float* lang, const uint8_t* lmask, void* prompt, uint8_t* pmask, int B, int S,
                            int extra, int C, hipStream_t s);
int esam3_launch_copy_rows(int dtype, const void* src, int n_src, void* dst, int n_dst, int dst_row0, int B, int C,
                           hipStream_t s);
int esam3_launch_bcast_rows(int dtype, const float* src, int n, void* dst, int n_dst, int dst_row0, int B, int C,
                            hipStream_t s);
int esam3_launch_box_sine(int dtype, const float* boxes, void* out, int64_t rows, int rows_per_img, hipStream_t s);
int esam3_launch_geo_tokens(int dtype, const float* points, const int32_t* plabels, const uint8_t* pmask, int Np,
                            const float* boxes, const int32_t* blabels, const uint8_t* bmask, int Nb, const void* imgn, int H,
                            int W, const float* w_pd, const float* b_pt, const float* w_bd, const float* b_bx,
                            const float* label_embed, const float* cls, void* x0, void* a_samp, void* a_encp, void* a_roi,
                            void* a_encb, uint8_t* gmask, int ld_mask, int mask_off, uint8_t* gmask_dense, int B, hipStream_t s);
int esam3_launch_rpb_mlp(const float* boxes, const float* const* wx /*w1,b1,w2,b2*/, const float* const* wy, float* out_y,
                         float* out_x, int64_t nq_total, int nq_img, int H, int W, int heads, hipStream_t s);
int esam3_launch_box_refine(int dtype, const void* delta, int ld, float* ref, int64_t rows, hipStream_t s);
int esam3_launch_masked_mean(int dtype, const void* x, const uint8_t* mask, void* out, int B, int S, int C, hipStream_t s);
int esam3_launch_dot_score(int dtype, const void* hs, int rows_per_img, int row0, int nq, const void* pp, float* out, int B,
                           int C, float scale, float clampv, hipStream_t s);
int esam3_launch_upsample_add(int dtype, const void* fine, const void* coarse, void* out_padded, int B, int h, int w, int C,
                              hipStream_t s);
int64_t esam3_groupnorm_scratch_floats(int B, int groups);
int esam3_launch_groupnorm_relu(int dtype, void* x, float* partial, const float* gamma, const float* beta, int B, int HW,
                                int C, int groups, float eps, hipStream_t s);
// ---- COCO RLE of binary masks (kernels_rle.hip)
int esam3_launch_rle_encode(const uint8_t* masks, int n, int H, int W, uint32_t* counts, int64_t capacity, int32_t* offsets,
                            void* scratch, hipStream_t s);
