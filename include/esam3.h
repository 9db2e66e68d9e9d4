/* esam3.h -- C ABI of the MI355X-native EfficientSAM3 image hot path (libesam3_hip.so).
 *
 * The reference is pure Python/PyTorch, so it has no FFI of its own; these entry points
 * are what a binding for the hot path would call, one per reference call site:
 *
 *   esam3_create / esam3_load_weight / esam3_finalize
 *       <- sam3/sam3/model_builder.py:944-1053 build_efficientsam3_image_model +
 *          _load_checkpoint (:584-630): one call per state_dict entry, reference key names.
 *   esam3_encode_image
 *       <- SAM3VLBackbone.forward_image (sam3/sam3/model/vl_combiner.py:81-124) plus the
 *          conv_s0/conv_s1 projection Sam3Processor.set_image[_batch] applies
 *          (sam3/sam3/model/sam3_image_processor.py:62-75,98-112).
 *   esam3_decode
 *       <- SAM3InteractiveImagePredictor._predict up to the mask decoder
 *          (sam3/sam3/model/sam1_task_predictor.py:328-421): prompt encoder
 *          (sam/prompt_encoder.py:155-197) + MaskDecoder.forward (sam/mask_decoder.py:107-163).
 *   esam3_postprocess_masks
 *       <- SAM2Transforms.postprocess_masks (sam3/sam3/model/utils/sam1_utils.py:77-119) and
 *          the `> mask_threshold` of sam1_task_predictor.py:423-428.
 *   esam3_encode_text (+ esam3_set_text_causal)
 *       <- TextStudentEncoder.forward after tokenisation (sam3/sam3/model/text_encoder_student.py:40-58).
 *   esam3_ground
 *       <- Sam3Image.forward_grounding (sam3/sam3/model/sam3_image.py:442-493): geometry encoder, fusion
 *          encoder, DETR decoder, scoring, mask head.
 *   esam3_rle_encode / esam3_rle_to_string / esam3_rle_from_string
 *       <- the evaluation writers' mask -> COCO RLE step (sam3/sam3/train/masks_ops.py:161-250).
 *   esam3_act_forward / _backward, esam3_linear_wgrad, esam3_dwconv_wgrad
 *       <- autograd's backward of the Conv2d / activation layers of a student ConvLayer (backbones/efficientvit/nn/ops.py:39-81)
 *   esam3_bn_train_forward / esam3_bn_train_backward
 *       <- nn.BatchNorm2d in training mode inside every student ConvLayer (backbones/efficientvit/nn/ops.py:69-77) and its backward
 *   esam3_stage1_update
 *       <- NativeScalerWithGradNormCount.__call__ (stage1/utils.py:347-362) + torch.optim.AdamW built by stage1/optimizer.py:6-46
 *   esam3_stage1_preprocess_u8 / esam3_stage1_preprocess_shape
 *       <- SA1BDataset.__getitem__'s image path (stage1/data/sa1b_dataset.py:163,170-171,217-228, transforms.py:48-88)
 *   esam3_distill_loss / esam3_distill_loss_backward
 *       <- masked_mse / masked_cosine_loss of stage-1 distillation (stage1/train_image_encoder_stage1.py:271-307) and the
 *          gradient of their weighted sum with respect to the student embedding (what loss.backward() hands to the trunk).
 *
 * Conventions: every pointer named *_dev is device memory owned by the caller; tensors are
 * NHWC (channels innermost) in the engine's activation dtype unless stated; all functions
 * return 0 on success and -1 on error (message via esam3_last_error()); no function
 * synchronises the stream; one engine per device, not thread-safe per engine.
 */
#ifndef ESAM3_H
#define ESAM3_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct esam3_engine esam3_engine;

enum { ESAM3_F32 = 0, ESAM3_BF16 = 1 };
enum { ESAM3_BACKBONE_EFFICIENTVIT = 0, ESAM3_BACKBONE_REPVIT = 1, ESAM3_BACKBONE_TINYVIT = 2,
       ESAM3_BACKBONE_VIT = 3 /* ViT-H teacher of build_sam3_image_model (model_builder.py:70-97) */ };

typedef struct esam3_config {
  int dtype;            /* ESAM3_F32 (validation) or ESAM3_BF16 (throughput) activations */
  int backbone;         /* ESAM3_BACKBONE_* */
  char model_name[16];  /* EfficientViT "b0" | "b1" | "b2"; RepViT "m0.9" | "m1.1"; TinyViT "5m" | "11m" | "21m"; ViT "vit_h" */
  int device;           /* HIP device ordinal */
  int interactive;      /* 1: sam2 neck + SAM heads are present (enable_inst_interactivity) */
  int fuse_linear_chains; /* 1: compose ConvT->1x1 and 3x3->conv_s0/s1 weight chains at finalize
                           (exact algebra, fewer FLOPs); 0: run the reference's layer list */
} esam3_config;

/* Outputs of the image encoder; NULL members are skipped.  B = batch.
 *   sam3_fpn: [B,288,288,256] [B,144,144,256] [B,72,72,256]   (all NULL -> sam3 neck not run)
 *   sam2_fpn: [B,288,288,32]  [B,144,144,64]  [B,72,72,256]   (after conv_s0 / conv_s1)
 *   trunk:    [B,72,72,1024]  ImageStudentEncoder output
 *   stages:   backbone stage outputs (validation taps): EfficientViT stage0..4, RepViT stage0..3,
 *             TinyViT patch embed + the 4 layer outputs, ViT-H ln_pre + the 4 global-block outputs */
typedef struct esam3_image_features {
  void* sam3_fpn_dev[3];
  void* sam2_fpn_dev[3];
  void* trunk_dev;
  void* stages_dev[5];
} esam3_image_features;

typedef struct esam3_prompts {
  const void* sam2_fpn_dev[3];  /* features of n_images images, as produced by esam3_encode_image */
  int n_images;
  int n_prompts;                /* Bp: independent prompt sets */
  const int32_t* prompt_image_dev; /* [Bp] image index of each prompt set */
  const float* coords_dev;      /* [Bp][n_points][2] (x, y) in network pixels (0..1008), may be NULL */
  const int32_t* labels_dev;    /* [Bp][n_points]: 0 neg, 1 pos, 2 box TL, 3 box BR, -1 pad */
  int n_points;
  const float* mask_input_dev;  /* optional [Bp][288][288] fp32 low-res mask logits */
  int multimask_output;
} esam3_prompts;

typedef struct esam3_decode_out {
  float* low_res_dev;   /* [Bp][C][288][288] fp32 logits (unclamped), C = 3 if multimask else 1 */
  float* iou_dev;       /* [Bp][C] */
  float* obj_score_dev; /* [Bp] object score logits, may be NULL */
} esam3_decode_out;

const char* esam3_last_error(void);
int esam3_create(const esam3_config* cfg, esam3_engine** out);
void esam3_destroy(esam3_engine* e);
/* host fp32 tensor with the reference's state_dict key; unknown keys are ignored */
int esam3_load_weight(esam3_engine* e, const char* name, const float* host_data,
                      const int64_t* shape, int ndim);
/* fold BatchNorm, pack to the GEMM layouts, upload; must precede encode/decode */
int esam3_finalize(esam3_engine* e);
/* Optional, after esam3_finalize: free the engine's fp32 host copies of the image-encoder and mask-decoder weights that
 * have been packed for the device and that a complete encode + decode pass never looks at again (esam3_load_weight keeps
 * one copy per tensor because layers are packed on first use).  Text / grounding weights are kept.  Returns the number of
 * bytes freed, -1 on error. */
int64_t esam3_release_host_weights(esam3_engine* e);

int esam3_encode_image(esam3_engine* e, const float* img_nchw_f32_dev, int B,
                       const esam3_image_features* out, void* hip_stream);
int esam3_decode(esam3_engine* e, const esam3_prompts* prompts, const esam3_decode_out* out,
                 void* hip_stream);
/* TextStudentEncoder.forward after tokenisation (text_encoder_student.py:40-58; MobileCLIP-S0 weights
 * under "backbone.language_backbone."): tokens int64 [B][S] (device) ->
 * memory fp32 [S][B][256] (language_features), embeds fp32 [S][B][dim] (language_embeds, dim = 512 or 768,
 * may be NULL).
 * The padding mask is tokens == 0 and stays on the host side. */
int esam3_encode_text(esam3_engine* e, const int64_t* tokens_dev, int B, int S, float* memory_sbd_dev,
                      float* embeds_sbd_dev, void* hip_stream);
/* cfg["causal_masking"] of the text student (model_builder.py:532-539: only MobileCLIP-B sets it): the text
 * self-attention then sees keys <= query (mobile_clip.py:826-846).  The variant ("mct" with RepMixer blocks or
 * "base"), depth and width are read from the state dict. */
int esam3_set_text_causal(esam3_engine* e, int causal_masking);
/* PCS grounding detector: Sam3Image.forward_grounding (sam3_image.py:442-493) for one text prompt per image plus
 * an optional geometric prompt (points / boxes; none = the dummy prompt), i.e. what Sam3Processor.set_text_prompt,
 * add_geometric_prompt and add_point_prompt run (sam3_image_processor.py:115-190,219-226; geometry encoder
 * geometry_encoders.py:600-695,732-853).  Weights: "geometry_encoder.*", "transformer.*",
 * "segmentation_head.*", "dot_prod_scoring.*". */
typedef struct esam3_ground_in {
  const void* sam3_fpn_dev[3];          /* sam3 neck levels of n_images images, as written by esam3_encode_image */
  int n_images;
  const float* language_features_dev;   /* [S][n_images][256] fp32 (esam3_encode_text layout) */
  const uint8_t* language_mask_dev;     /* [n_images][S], 1 = padding token */
  int n_tokens;                         /* S */
  /* geometric prompt; per image the valid entries come first (right-padded), masks may be NULL = all valid */
  int n_points;                         /* Np = max points per image, 0 = none */
  const float* points_dev;              /* [n_images][Np][2] x, y normalised to [0, 1] */
  const int32_t* point_labels_dev;      /* [n_images][Np] 1 = positive, 0 = negative */
  const uint8_t* point_mask_dev;        /* [n_images][Np] 1 = padding */
  int n_boxes;                          /* Nb = max boxes per image, 0 = none */
  const float* boxes_dev;               /* [n_images][Nb][4] cx, cy, w, h normalised to [0, 1] */
  const int32_t* box_labels_dev;        /* [n_images][Nb] 1 = positive, 0 = negative */
  const uint8_t* box_mask_dev;          /* [n_images][Nb] 1 = padding */
} esam3_ground_in;
typedef struct esam3_ground_out {
  float* pred_logits_dev;     /* [n_images][200] */
  float* pred_boxes_dev;      /* [n_images][200][4] cx, cy, w, h in [0, 1] */
  float* presence_logit_dev;  /* [n_images] */
  float* pred_masks_dev;      /* [n_images][200][288][288] fp32 logits */
  float* semantic_seg_dev;    /* [n_images][288][288] fp32 logits, may be NULL */
} esam3_ground_out;
int esam3_ground(esam3_engine* e, const esam3_ground_in* in, const esam3_ground_out* out, void* hip_stream);
/* low_res: [n][288][288] fp32 -> masks at (out_h, out_w); either output may be NULL */
int esam3_postprocess_masks(esam3_engine* e, const float* low_res_dev, int n_masks, int out_h,
                            int out_w, float max_hole_area, float mask_threshold,
                            uint8_t* masks_u8_dev, float* masks_logits_dev, void* hip_stream);
/* in-place clamp of fp32 logits (sam1_task_predictor.py:426) */
int esam3_clamp_f32(esam3_engine* e, float* x_dev, int64_t n, float lo, float hi, void* hip_stream);

/* Sam3Processor.transform for an already network-sized image (sam3_image_processor.py:24-31):
 * uint8 HWC [B][H][W][3] -> fp32 NCHW [B][3][H][W], x/255 then (x-0.5)/0.5 */
int esam3_preprocess_u8(const uint8_t* img_hwc_u8_dev, float* out_nchw_f32_dev, int B, int H, int W,
                        void* hip_stream);
/* Sam3Processor.transform for an image of any size (sam3_image_processor.py:24-31,57-58:
 * v2.Resize on a uint8 device tensor = fp32 antialiased bilinear (triangle filter, support
 * max(scale,1)), round half to even, back to uint8; then /255 and (x-0.5)/0.5):
 * uint8 HWC [H][W][3] -> fp32 CHW [3][out_h][out_w] */
int esam3_preprocess_resize_u8(const uint8_t* img_hwc_u8_dev, int H, int W, float* out_chw_f32_dev,
                               int out_h, int out_w, void* hip_stream);
/* the same for B images of one size in ONE launch (Sam3Processor.set_image_batch, sam3_image_processor.py:86-113,
 * applies the transform per image in a Python loop): uint8 [B][H][W][3] -> fp32 [B][3][out_h][out_w] */
int esam3_preprocess_resize_u8_batch(const uint8_t* imgs_bhwc_u8_dev, int B, int H, int W, float* out_bchw_f32_dev,
                                     int out_h, int out_w, void* hip_stream);
/* the same from 4-byte pixels, uint8 [B][H][W][4] = (R, G, B, ignored): the in-memory layout of a PIL "RGB" image, which
 * Pillow exports without a copy (Image.__arrow_c_array__); packing it to 3 bytes on the host (Image.tobytes, 3 ms per
 * 1024 x 1024 image) is what bounds the reference-shaped set_image_batch call path.  Same arithmetic, same output. */
int esam3_preprocess_resize_rgbx_batch(const uint8_t* imgs_bhw4_u8_dev, int B, int H, int W, float* out_bchw_f32_dev,
                                       int out_h, int out_w, void* hip_stream);

/* COCO run-length encoding of binary masks, the mask -> RLE step of the evaluation writers
 * (scripts/eval/gold/eval_efficientsam3_all_subsets.py:124-135 through pycocotools.mask.encode;
 * sam3/sam3/train/masks_ops.py:161-230 rle_encode): masks u8 [n][H][W] (non-zero = foreground) -> run lengths in
 * column-major order (zeros, ones, zeros, ...; the first is 0 when a mask starts with a one) of all masks back to
 * back in counts_dev [capacity] and offsets_dev [n+1] (mask i owns counts[offsets[i] .. offsets[i+1])).  No hidden
 * sync: when offsets[n] > capacity the counts are truncated and the caller retries with a larger buffer.
 * scratch_dev: esam3_rle_scratch_bytes(n, H, W, capacity) bytes. */
int64_t esam3_rle_scratch_bytes(int n, int H, int W, int64_t capacity);
int esam3_rle_encode(const uint8_t* masks_dev, int n, int H, int W, uint32_t* counts_dev, int64_t capacity,
                     int32_t* offsets_dev, void* scratch_dev, int64_t scratch_bytes, void* hip_stream);
/* cocoapi's compressed "counts" string (maskApi.c rleToString / rleFrString; what pycocotools returns as bytes):
 * host buffers, returns the number of characters / counts written, -1 on error */
int64_t esam3_rle_to_string(const uint32_t* counts_host, int64_t n_counts, char* out, int64_t capacity);
/* dst_host[i] = (float) src_host[i], i < n: the uint8 -> float32 widening of thresholded masks on the host (the reference returns
 * float32 numpy masks, sam3/model/sam1_task_predictor.py:293-295); thread-safe on disjoint ranges */
int esam3_host_widen_u8_f32(const uint8_t* src_host, float* dst_host, int64_t n);
int64_t esam3_rle_from_string(const char* s, int64_t len, uint32_t* counts_host, int64_t capacity);

/* Stage-1 distillation loss, forward (stage1/train_image_encoder_stage1.py:271-307 masked_mse and
 * masked_cosine_loss): student embedding preds and teacher embedding, both token-major [B][HW][C] (NHWC; C % 8 == 0;
 * dtype 0 = fp32, 1 = bf16, teacher also 2 = fp16 as stored by save_embedding_image_stage1.py:92-96), valid [B][HW]
 * u8 from build_valid_mask.  per_image_dev [B][2] = {sum_valid sum_c (p-t)^2 / max(#valid,1),
 * sum_valid (1 - cos(p,t)) / max(#valid,1)}; the losses are the means of the two columns.
 * scratch_dev: B*HW*2 floats. */
int esam3_distill_loss(int preds_dtype, const void* preds_dev, int teacher_dtype, const void* teacher_dev,
                       const uint8_t* valid_dev, int B, int HW, int C, float* per_image_dev, float* scratch_dev,
                       void* hip_stream);

/* dL/dpreds of  masked_mse + cosine_weight * masked_cosine_loss  (the loss of stage1/train_image_encoder_stage1.py:186-210),
 * times grad_scale (1 / ACCUMULATION_STEPS): grad_preds_dev [B][HW][C] in the dtype of preds, zero at masked pixels.
 * scratch_dev: B floats. */
int esam3_distill_loss_backward(int preds_dtype, const void* preds_dev, int teacher_dtype, const void* teacher_dev,
                                const uint8_t* valid_dev, int B, int HW, int C, float cosine_weight, float grad_scale,
                                void* grad_preds_dev, float* scratch_dev, void* hip_stream);
/* The same gradient times a scale held in DEVICE memory (the AMP loss scale inside the updater's state; a power of two, so exact wherever it
 * multiplies): a training step that calls this form never reads the scale back to the host (that read was the step's only synchronisation). */
int esam3_distill_loss_backward_ds(int preds_dtype, const void* preds_dev, int teacher_dtype, const void* teacher_dev,
                                   const uint8_t* valid_dev, int B, int HW, int C, float cosine_weight, float grad_scale,
                                   const float* scale_dev, void* grad_preds_dev, float* scratch_dev, void* hip_stream);

/* Gradient kernels of the layers an EfficientViT ConvLayer / DSConv / MBConv is made of (backbones/efficientvit/nn/ops.py:39-81,
 * 264-360: Conv2d without bias -> BatchNorm2d -> activation), NHWC rows, dtype 0 fp32 / 1 bf16 activations and gradients, fp32
 * weight gradients.  Building blocks of the student-trunk backward of stage 1 (stage1/train_image_encoder_stage1.py:196-217):
 * efficientsam3_amd/stage1_train.py sequences them into the training step of every student; tests/test_train_blocks.py composes them into an
 * MBConv block and checks every gradient against autograd.
 *   esam3_act_forward / _backward: y = act(x), dx = dy * act'(x); act 0 none, 1 ReLU, 2 GELU (erf), 3 Hardswish, 4 sigmoid
 *     (the SqueezeExcite gate of the RepViT students); n % 8 == 0.
 *   esam3_linear_wgrad: dw[N][K] = sum_rows dy[row][n] * x[row][k] (the weight gradient of a 1x1 conv / Linear with weight
 *     [N][K]; the M rows are the reduction: split over the rows, fp32 partial tiles summed in a fixed order), dbias[N] = sum_rows dy
 *     (NULL to skip).  N % 8 == 0, K % 8 == 0.  workspace: esam3_linear_wgrad_workspace(M, N, K) bytes.
 *   esam3_dwconv_wgrad: dw[C][1][k][k] of a depthwise k x k (3 | 5), padding k / 2, stride 1 | 2: x [B][H][W][C], dy [B][ceil(H/s)][ceil(W/s)][C].
 *   esam3_dwconv_dgrad: its data gradient (a transposed convolution for stride 2).
 * The data gradient of a 1x1 conv reuses the forward operator: esam3_op_linear with the transposed weight. */
int esam3_act_forward(int dtype, const void* x_dev, void* y_dev, int64_t n, int act, void* hip_stream);
int esam3_act_backward(int dtype, const void* x_dev, const void* dy_dev, void* dx_dev, int64_t n, int act, void* hip_stream);
int64_t esam3_linear_wgrad_workspace(int64_t M, int N, int K);
int esam3_linear_wgrad(int dtype, const void* dy_dev, const void* x_dev, int64_t M, int N, int K, float* dw_dev, float* dbias_dev,
                       void* workspace_dev, void* hip_stream);
/* dw [Cout][Cin][3][3] fp32 of a dense 3x3 conv with padding 1 and stride 1 | 2 (the student head's second conv, stage1/model.py:197-200; the
 * second conv of the RepViT / TinyViT patch embedding, repvit.py:230, tiny_vit.py:80) in ONE launch of the weight-gradient GEMM: grid.z = 9 taps x
 * row splits, the x operand of tap (ky, kx) gathered from the input pixel (oy s + ky - 1, ox s + kx - 1), zeros outside the image.
 * x [B][IH][IW][Cin], dy [B][ceil(IH/s)][ceil(IW/s)][Cout]; Cin % 8 == 0, Cout % 8 == 0; workspace esam3_conv3x3_wgrad_workspace(...) bytes. */
int64_t esam3_conv3x3_wgrad_workspace(int B, int IH, int IW, int Cin, int Cout, int stride);
int esam3_conv3x3_wgrad(int dtype, const void* dy_dev, const void* x_dev, int B, int IH, int IW, int Cin, int Cout, int stride, float* dw_dev,
                        void* workspace_dev, void* hip_stream);
/* out[N] = sum over the M rows of dy[M][N] (the bias gradient of a conv / Linear whose BatchNorm is absent: the local MBConv of an
 * EfficientViTBlock, ops.py:704-711 use_bias=(True, True, False), norm=(None, None, bn2d)); workspace: esam3_colsum_workspace(M, N) bytes */
int64_t esam3_colsum_workspace(int64_t M, int N);
int esam3_colsum(int dtype, const void* dy_dev, int64_t M, int N, float* out_dev, void* workspace_dev, void* hip_stream);
/* Per-channel scale / shift and per-image channel reductions: the elementwise and reduction halves of the two RepViT layers that are not
 * plain convolutions, forwards and backwards (round 5: RepViT students of stage 1, stage1/model.py:386-395) --
 * RepVGGDW (sam3/backbones/repvit.py:84-93: bn(conv_bn(x) + conv1(x) + x), conv1 a depthwise 1x1 with bias) and timm's SqueezeExcite
 * (repvit.py:136,150: x * sigmoid(fc2(relu(fc1(mean_hw(x)))))).  x / add / out [B][HW][C] in dtype, mul / bias / out of the reduction fp32.
 *   esam3_channel_scale: out = add + x * (mul[(b,) c] + plus_one) + bias[(b,) c] * bias_scale; mul / bias are [C], or [B][C] when
 *     mul_per_image / bias_per_image; add and bias may be NULL.  C % 8 == 0.
 *   esam3_batched_coldot: out[b][c] = scale * sum over the HW pixels of a[b][p][c] * (b2 ? b2[b][p][c] : 1): the SE mean (b2 NULL,
 *     scale 1 / HW), the gate's gradient (sum dy x), and with B = 1 the depthwise-1x1 weight gradient (sum over all rows of ds x).
 *     Fixed summation order (deterministic).  C % 8 == 0, C <= 2048; workspace: esam3_batched_coldot_workspace(B, C) bytes. */
int esam3_channel_scale(int dtype, const void* x_dev, const float* mul_dev, int mul_per_image, float plus_one, const float* bias_dev,
                        int bias_per_image, float bias_scale, const void* add_dev, void* out_dev, int B, int64_t HW, int C, void* hip_stream);
int64_t esam3_batched_coldot_workspace(int B, int C);
int esam3_batched_coldot(int dtype, const void* a_dev, const void* b2_dev, int B, int64_t HW, int C, float scale, float* out_dev,
                         void* workspace_dev, void* hip_stream);
int64_t esam3_dwconv_wgrad_workspace(int C);
int esam3_dwconv_wgrad(int dtype, const void* x_dev, const void* dy_dev, int B, int H, int W, int C, int ksize, int stride,
                       float* dw_dev, void* workspace_dev, void* hip_stream);
/* Backward of LiteMLA's ReLU linear attention (backbones/efficientvit/nn/ops.py:584-621 relu_linear_att): ms [B][N][groups*3*dim] is
 * the multi-scale qkv tensor (per head group the channels [q dim | k dim | v dim]), dout [B][N][groups*dim] the gradient of the
 * attention output; dms receives d(ms) (the ReLU masks of q and k applied); y_dev (may be NULL) receives the forward output.
 * dim 16 or 32; fp32 arithmetic; one workgroup per (image, head group), deterministic. */
int esam3_lite_mla_backward(int dtype, const void* ms_dev, const void* dout_dev, void* dms_dev, void* y_dev, int B, int N, int groups,
                            int dim, float eps, void* hip_stream);
/* the same with the tokens split over several workgroups per (image, head group) and the two small matrices summed in a fixed order
 * (round 5: 128 - 256 workgroups walking up to 3969 tokens three times were 24 % of a training step); workspace_dev holds
 * esam3_lite_mla_backward_workspace(B, N, groups, dim) bytes; results as esam3_lite_mla_backward up to the summation order of S / dS.
 * dout_dev == NULL: forward only -- y_dev receives relu_linear_att(ms), dms_dev is not touched (the training-mode forward of LiteMLA) */
int64_t esam3_lite_mla_backward_workspace(int B, int N, int groups, int dim);
int esam3_lite_mla_backward_ws(int dtype, const void* ms_dev, const void* dout_dev, void* dms_dev, void* y_dev, int B, int N, int groups,
                               int dim, float eps, void* workspace_dev, void* hip_stream);
/* round 6: the same with the multi-scale tensor as its two halves -- ms0_dev = the qkv conv's output, ms1_dev = the aggregated scale, each
 * [B][N][groups / 2 * 3 * dim]; gradients leave as dms0_dev / dms1_dev (no torch.cat in front, no slice copies behind); N > 256 tokens */
int esam3_lite_mla_backward_ws2(int dtype, const void* ms0_dev, const void* ms1_dev, const void* dout_dev, void* dms0_dev, void* dms1_dev,
                                void* y_dev, int B, int N, int groups, int dim, float eps, void* workspace_dev, void* hip_stream);
/* dx [B][H][W][C] of the same depthwise conv from dy [B][ceil(H/s)][ceil(W/s)][C]; w_dev fp32 [C][1][k][k] ON THE DEVICE */
int esam3_dwconv_dgrad(int dtype, const void* dy_dev, const float* w_dev, void* dx_dev, int B, int H, int W, int C, int ksize,
                       int stride, void* hip_stream);

/* Training-path operators on DEVICE-RESIDENT fp32 master weights (csrc/kernels_train_dev.hip): the parameters of a stage-1 student
 * live in the optimizer's flat fp32 arena on the device (esam3_stage1_update) and change every step, so these entry points take the
 * weight where it lies, re-pack it on the device into the forward kernel's layout (workspace_dev) and launch the engine's own
 * kernels -- no host copy, no synchronisation.  Replaces, for the student of stage1/model.py:188-211 under model.train()
 * (stage1/train_image_encoder_stage1.py:165-217), the forward of every Conv2d and, with `transpose` / `dgrad`, its data gradient.
 *   esam3_train_pack_bytes: workspace bytes of esam3_train_linear (N, K) / esam3_train_conv3x3 (N = Cout, K = 9 Cin);
 *     esam3_train_dwconv needs 4 k k C bytes, esam3_train_stem 108 Cout bytes.
 *   esam3_train_linear: out[M][N] = x[M][K] . W^T + bias, W = w_dev [N][K] (transpose = 0) or w_dev [K][N] read transposed
 *     (transpose = 1: the data gradient dx[M][N] = dy[M][K] . W of a Linear whose weight is [K][N]).
 *   esam3_train_conv3x3: 3x3, padding 1, w_dev [Cout][Cin][3][3] (+ bias); dgrad = 1: x is dy [B][H][W][Cin] with Cin = the
 *     FORWARD conv's output channels, w_dev the forward weight [Cin][Cout][3][3], out = dx [B][H][W][Cout].
 *   esam3_train_dwconv: depthwise k x k (3 | 5), stride 1 | 2, w_dev [C][1][k][k].  esam3_train_stem: the 3 -> Cout stride-2 3x3
 *     stem conv on the fp32 NCHW image, w_dev [Cout][3][3][3], no bias / activation (BatchNorm + Hardswish follow as steps).
 *   esam3_resize_bilinear_backward: dx [B][IH][IW][C] = adjoint of F.interpolate(x, (OH, OW), bilinear, align_corners=False)
 *     applied to dy [B][OH][OW][C] (the student's final resize, stage1/model.py:205-210); C % 8 == 0. */
int64_t esam3_train_pack_bytes(int dtype, int N, int K);
int esam3_train_linear(int dtype, const void* x_dev, const float* w_dev, const float* bias_dev, void* out_dev, int64_t M, int N, int K,
                       int transpose, void* workspace_dev, void* hip_stream);
int esam3_train_conv3x3(int dtype, const void* x_dev, const float* w_dev, const float* bias_dev, void* out_dev, int B, int H, int W,
                        int Cin, int Cout, int dgrad, void* workspace_dev, void* hip_stream);
/* the same conv through the bordered-input form of the implicit GEMM (round 6: the engine's 256 x 256 tile kernel on a bordered copy of the
 * input kept in the workspace; bf16, Cin % 64 == 0, Cout >= 192, B H W % 256 == 0 -- any other shape, or a workspace smaller than
 * esam3_train_conv3x3_workspace bytes, runs esam3_train_conv3x3).  Reference: the student head's 3x3 conv, stage1/model.py:386-420 */
int64_t esam3_train_conv3x3_workspace(int dtype, int B, int H, int W, int Cin, int Cout);
int esam3_train_conv3x3_ws(int dtype, const void* x_dev, const float* w_dev, const float* bias_dev, void* out_dev, int B, int H, int W,
                           int Cin, int Cout, int dgrad, void* workspace_dev, int64_t workspace_bytes, void* hip_stream);
/* the dense 3x3 (padding 1) with STRIDE 2: the second conv of RepViT's patch embedding (sam3/backbones/repvit.py:229-230), forward only
 * -- out [B][ceil(H/2)][ceil(W/2)][Cout]; workspace esam3_train_pack_bytes(dtype, Cout, 9 Cin).  Its data gradient is
 * esam3_train_conv3x3(dgrad = 1) on dy spread over the even pixels of a zero H x W grid, its weight gradient esam3_linear_wgrad per tap. */
int esam3_train_conv3x3_s2(int dtype, const void* x_dev, const float* w_dev, const float* bias_dev, void* out_dev, int B, int H, int W,
                           int Cin, int Cout, void* workspace_dev, void* hip_stream);
int esam3_train_dwconv(int dtype, const void* x_dev, const float* w_dev, const float* bias_dev, void* out_dev, int B, int H, int W, int C,
                       int ksize, int stride, void* workspace_dev, void* hip_stream);
/* dx [B][H][W][C] of the depthwise conv above from dy [B][ceil(H/s)][ceil(W/s)][C] (autograd's conv backward for the student's
 * depthwise layers, stage1/train_image_encoder_stage1.py:186-217): stride 1 = the forward kernels on dy with the rotated kernel
 * (packed into workspace_dev, 4 k k C bytes); stride 2 = esam3_dwconv_dgrad */
int esam3_train_dwconv_dgrad(int dtype, const void* dy_dev, const float* w_dev, void* dx_dev, int B, int H, int W, int C, int ksize,
                             int stride, void* workspace_dev, void* hip_stream);
/* the `x` operand of the stem's weight gradient: im2col of the 3 -> C0 3x3 / stride 2 / padding 1 stem conv (backbone.py:50-58) on the NCHW
 * fp32 image -> out_dev [B * ceil(H/2) * ceil(W/2)][32] in the activation dtype, column c * 9 + kh * 3 + kw (torch.nn.functional.unfold's
 * order), columns 27..31 zero; dW = esam3_linear_wgrad(d_conv, out)[:, :27] */
int esam3_stem_im2col(int dtype, const float* img_nchw_dev, void* out_dev, int B, int H, int W, void* hip_stream);
int esam3_train_stem(int dtype, const float* img_nchw_dev, const float* w_dev, void* out_dev, int B, int H, int W, int Cout,
                     void* workspace_dev, void* hip_stream);
int esam3_resize_bilinear_backward(int dtype, const void* dy_dev, void* dx_dev, int B, int IH, int IW, int OH, int OW, int C,
                                   void* hip_stream);

/* What a TinyViTBlock needs in training mode beyond the kernels above (csrc/kernels_train_vit.hip; round 5: the TinyViT students of stage 1,
 * stage1/model.py:397-406 -> TinyViTAdapter over sam3/backbones/tiny_vit.py).  fp32 arithmetic, fp32 / bf16 rows.
 *   esam3_ln_train_forward / _backward: nn.LayerNorm over the C channels of [M][C] rows (Attention.norm, Mlp.norm: tiny_vit.py:201,236) with
 *     the saved per-row mean / rstd; backward = (dx, dgamma, dbeta), the two column sums over fixed row partitions (deterministic).
 *     C % 8 == 0, C <= 1024; workspace esam3_ln_train_workspace(C) bytes.
 *   esam3_win_attn_train_forward / _backward: the attention of Attention.forward between its two Linear layers (tiny_vit.py:271-293), per window
 *     and head: out = softmax(q k^T scale + bias) v on qkv [windows][N][heads * 96] (per head q | k | v, 32 channels each), bias
 *     [heads][N][N] fp32 = attention_biases[:, attention_bias_idxs] (symmetric in (i, j)), out [windows][N][heads * 32], lse
 *     [windows][heads][N] fp32 (log-sum-exp of every row, kept for the backward).  Backward: dqkv like qkv, and ds
 *     [windows][heads][N][N] fp32 = the gradient of the bias-added logits -- summed over the windows (esam3_colsum) it is the gradient of
 *     `bias`.  N <= 256.
 *   esam3_attn_bias_gather_sum: out[h][o] = sum of full[h][item] over the items (flattened (i, j)) whose attention_bias_idxs is o, given as
 *     CSR lists (start [n_off + 1], items [NN]) on the device: the gradient of attention_biases [heads][n_off] from that of the gathered table. */
int esam3_ln_train_forward(int dtype, const void* x_dev, void* y_dev, int64_t M, int C, const float* gamma_dev, const float* beta_dev, float eps,
                           float* mean_dev, float* rstd_dev, void* hip_stream);
int64_t esam3_ln_train_workspace(int C);
int esam3_ln_train_backward(int dtype, const void* x_dev, const void* dy_dev, const float* gamma_dev, const float* mean_dev, const float* rstd_dev,
                            void* dx_dev, float* dgamma_dev, float* dbeta_dev, int64_t M, int C, void* workspace_dev, void* hip_stream);
int esam3_win_attn_train_forward(int dtype, const void* qkv_dev, const float* bias_dev, void* out_dev, float* lse_dev, int windows, int N, int heads,
                                 float scale, void* hip_stream);
int esam3_win_attn_train_backward(int dtype, const void* qkv_dev, const float* bias_dev, const void* out_dev, const float* lse_dev,
                                  const void* dout_dev, void* dqkv_dev, float* ds_dev, int windows, int N, int heads, float scale, void* hip_stream);
/* The same two operations given the attention_biases parameter itself beside the gathered table: tab_dev [heads][ws * ws], N = ws * ws tokens,
 * bias[h][i][j] = tab[h][|dy| * ws + |dx|] (tiny_vit.py:240-254; ws <= 16).  The bf16 kernels (round 6: the products on the matrix unit, as the
 * reference's autocast runs them) index the head's row of `tab` from LDS; fp32 reads `bias_dev` as before. */
int esam3_win_attn_train_forward_tab(int dtype, const void* qkv_dev, const float* bias_dev, const float* tab_dev, int ws, void* out_dev, float* lse_dev,
                                     int windows, int heads, float scale, void* hip_stream);
int esam3_win_attn_train_backward_tab(int dtype, const void* qkv_dev, const float* bias_dev, const float* tab_dev, int ws, const void* out_dev,
                                      const float* lse_dev, const void* dout_dev, void* dqkv_dev, float* ds_dev, int windows, int heads, float scale,
                                      void* hip_stream);
int esam3_attn_bias_gather_sum(const float* full_dev, const int* start_dev, const int* items_dev, float* out_dev, int heads, int NN, int n_off,
                               void* hip_stream);
/* TinyViTBlock's window partition / reverse (tiny_vit.py:350-374) as one copy: x_dev [B][H][W][C] <-> windows_dev [B * ceil(H/ws) * ceil(W/ws)][ws*ws][C]
 * (zeros in the padded rows / columns; reverse = 1 drops them).  dtype 0 fp32 / 1 bf16; rows of a multiple of 16 bytes. */
int esam3_window_partition(int dtype, const void* src_dev, void* dst_dev, int B, int H, int W, int C, int ws, int reverse, void* hip_stream);

/* Update half of the stage-1 training step: AMP loss scaler + gradient-norm clipping + AdamW on ONE flat fp32 arena.
 * Replaces, for a student whose trainable parameters live in `params` (each tensor padded to a multiple of 256 elements),
 *   NativeScalerWithGradNormCount.__call__ after backward (stage1/utils.py:347-362): GradScaler.unscale_ (non-finite
 *     check + 1/scale), clip_grad_norm_(parameters, clip_grad) -- or ampscaler_get_grad_norm (:324-338) when clip_grad <= 0
 *     --, GradScaler.step(optimizer) (skipped when a gradient was inf / nan), GradScaler.update() (x growth_factor after
 *     growth_interval clean steps, x backoff_factor after a skipped one), called from train_one_epoch
 *     (stage1/train_image_encoder_stage1.py:210-217) together with optimizer.zero_grad() (:218-219, `zero_grads`);
 *   torch.optim.AdamW as stage1/optimizer.py:6-29 builds it: decoupled weight decay only on the has_decay group of
 *     set_weight_decay (:32-46: chunk_decay[c] = 1), per-group lr = lr x lr_scale (utils.py:557-620; chunk_lr_scale[c]).
 * All device pointers; chunk tables have n / 256 entries (NULL: every chunk decays / scale 1).  state16 (16 floats, device)
 * carries [0] loss scale, [1] growth tracker, [2] found_inf of this call, [3] total gradient norm of this call (the value the
 * reference returns), [4] AdamW step count, [5] clip coefficient, [6..7] bias corrections; initialise it to
 * {init_scale (65536), 0, ...}.  amp_enabled = 0 is GradScaler(enabled=False): scale 1, never skipped.  bf16_params_out
 * (optional, n bf16) receives the updated weights in the engine's compute dtype.  No host synchronisation; returns 0 / -1.
 * The optimizer's hyper-parameters are doubles, as torch holds them (1 - beta2 is formed in double before it becomes an fp32
 * scalar).  workspace_dev: esam3_stage1_update_workspace(n) bytes. */
int64_t esam3_stage1_update_workspace(int64_t n);
int esam3_stage1_update(float* params_dev, float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                        const float* chunk_lr_scale_dev, const uint8_t* chunk_decay_dev, double lr, double beta1, double beta2,
                        double eps, double weight_decay, float clip_grad, float* state16_dev, float growth_factor,
                        float backoff_factor, int growth_interval, int amp_enabled, int zero_grads, void* bf16_params_out_dev,
                        void* workspace_dev, void* hip_stream);

/* BatchNorm2d in TRAINING mode on NHWC rows [rows = B H W][C] (dtype 0 fp32 / 1 bf16 activations and gradients; C % 8 == 0,
 * C <= 2048): what every ConvLayer's nn.BatchNorm2d does while stage 1 trains (backbones/efficientvit/nn/ops.py:69-77,
 * nn/norm.py:47; stage1/train_image_encoder_stage1.py:165 model.train(), :310-314) and its autograd backward.  Building blocks of
 * the student-trunk backward (efficientsam3_amd/train_blocks.py: ConvLayerTrain); the inference engine folds BatchNorm instead.
 * forward : y = (x - mean_c) / sqrt(var_c + eps) * gamma_c + beta_c with the batch mean and BIASED variance over the rows;
 *           running_mean / running_var (may be NULL) <- (1 - momentum) * old + momentum * (mean, UNBIASED variance);
 *           save_mean / save_rstd [C] fp32 for the backward.
 * backward: dbeta = sum dy, dgamma = sum dy * xhat, dx = gamma * rstd * (dy - dbeta / n - xhat * dgamma / n).
 * workspace_dev: esam3_bn_train_workspace(C) bytes (per-split partial sums: what a SyncBatchNorm would all-reduce). */
int64_t esam3_bn_train_workspace(int C);
int esam3_bn_train_forward(int dtype, const void* x_dev, void* y_dev, int64_t rows, int C, const float* gamma_dev,
                           const float* beta_dev, float* running_mean_dev, float* running_var_dev, double momentum, double eps,
                           float* save_mean_dev, float* save_rstd_dev, void* workspace_dev, void* hip_stream);
int esam3_bn_train_backward(int dtype, const void* x_dev, const void* dy_dev, void* dx_dev, int64_t rows, int C,
                            const float* gamma_dev, const float* save_mean_dev, const float* save_rstd_dev, float* dgamma_dev,
                            float* dbeta_dev, void* workspace_dev, void* hip_stream);
/* The ConvLayer's activation applied in the BatchNorm's own passes (Conv2d -> BatchNorm2d -> act, backbones/efficientvit/nn/ops.py:39-81 and the
 * Conv2d_BN + activation pairs of repvit.py / tiny_vit.py): forward writes y (kept for the backward; may be NULL) AND y_act = act(y); backward takes dy =
 * the gradient of act(y) and pre = y, and uses dy act'(pre) where esam3_bn_train_backward uses dy.  act 1 ReLU, 2 GELU, 3 Hardswish, 4 sigmoid. */
int esam3_bn_act_train_forward(int dtype, const void* x_dev, void* y_dev, void* y_act_dev, int act, int64_t rows, int C, const float* gamma_dev,
                               const float* beta_dev, float* running_mean_dev, float* running_var_dev, double momentum, double eps,
                               float* save_mean_dev, float* save_rstd_dev, void* workspace_dev, void* hip_stream);
int esam3_bn_act_train_backward(int dtype, const void* x_dev, const void* dy_dev, const void* pre_dev, int act, void* dx_dev, int64_t rows, int C,
                                const float* gamma_dev, const float* save_mean_dev, const float* save_rstd_dev, float* dgamma_dev,
                                float* dbeta_dev, void* workspace_dev, void* hip_stream);
/* The same backward without the BatchNorm's saved output: pre = y is recomputed from x, the saved statistics, gamma and beta exactly as the forward
 * formed and stored it (fp32 expression, rounded to the storage dtype), so esam3_bn_act_train_forward may be called with y_dev = NULL: two of the
 * seven tensor passes of the backward and one of the four of the forward are gone, and a layer keeps one tensor less between the two. */
int esam3_bn_act_train_backward_rc(int dtype, const void* x_dev, const void* dy_dev, int act, void* dx_dev, int64_t rows, int C,
                                   const float* gamma_dev, const float* beta_dev, const float* save_mean_dev, const float* save_rstd_dev,
                                   float* dgamma_dev, float* dbeta_dev, void* workspace_dev, void* hip_stream);
/* The same two operations in halves, for SyncBatchNorm (stage1/train_image_encoder_stage1.py:62-63 `--use-sync-bn`:
 * torch.nn.SyncBatchNorm.convert_sync_batchnorm): between the statistics and the elementwise map the host combines the ranks' values with ONE
 * collective each way, as torch's SyncBatchNorm does (efficientsam3_amd/train_blocks.py: bn_train_forward / bn_train_backward).
 *   esam3_bn_train_stats: this rank's mean, rstd = (var + eps)^-1/2 and biased variance per channel (no map, running statistics untouched);
 *   esam3_bn_train_apply: y = (x - mean) rstd gamma + beta with the (all-rank) mean / rstd given;
 *   esam3_bn_train_backward_sums: this rank's sum dy xhat (= its dgamma) and sum dy (= its dbeta) with the all-rank mean / rstd;
 *   esam3_bn_train_backward_apply: dx = gamma rstd (dy - sum_dy / n - xhat sum_dy_xhat / n) with the ALL-RANK sums and n = total_rows (pass
 *     total_rows = 1 with sums the host already divided by the all-rank row count: no read-back of the count is needed then). */
int esam3_bn_train_stats(int dtype, const void* x_dev, int64_t rows, int C, double eps, float* mean_dev, float* rstd_dev, float* var_dev,
                         void* workspace_dev, void* hip_stream);
int esam3_bn_train_apply(int dtype, const void* x_dev, void* y_dev, int64_t rows, int C, const float* gamma_dev, const float* beta_dev,
                         const float* mean_dev, const float* rstd_dev, void* hip_stream);
int esam3_bn_train_backward_sums(int dtype, const void* x_dev, const void* dy_dev, int64_t rows, int C, const float* mean_dev,
                                 const float* rstd_dev, float* sum_dy_xhat_dev, float* sum_dy_dev, void* workspace_dev, void* hip_stream);
int esam3_bn_train_backward_apply(int dtype, const void* x_dev, const void* dy_dev, void* dx_dev, int64_t rows, int C, const float* gamma_dev,
                                  const float* mean_dev, const float* rstd_dev, const float* sum_dy_xhat_dev, const float* sum_dy_dev,
                                  double total_rows, void* hip_stream);

/* Stage-1 input pipeline (BASELINE config 5): what SA1BDataset.__getitem__ does to an image before the trunks see it
 * (stage1/data/sa1b_dataset.py:163,170-171,217-228; stage1/data/transforms.py:48-55,81-88): ResizeLongestSide(img_size)
 * as an fp32 antialiased bilinear resize (no rounding back to uint8), (x - pixel_mean) / pixel_std with the three
 * per-channel host values (the reference's defaults 123.675, 116.28, 103.53 / 58.395, 57.12, 57.375), zero padding at
 * the bottom / right to img_size x img_size.  img: uint8 [H][W][3] on the device; out: fp32 [3][img_size][img_size].
 * The resized (un-padded) size is returned through new_h / new_w (host; `img_size_before_pad` of the dataset, the input
 * of build_valid_mask).  esam3_stage1_preprocess_shape is get_preprocess_shape alone (host arithmetic). */
void esam3_stage1_preprocess_shape(int H, int W, int img_size, int* new_h, int* new_w);
int esam3_stage1_preprocess_u8(const uint8_t* img_hwc_u8_dev, int H, int W, float* out_chw_f32_dev, int img_size,
                               const float* pixel_mean3_host, const float* pixel_std3_host, int* new_h, int* new_w,
                               void* hip_stream);

/* Profiler scopes.  The engine brackets the launches of each phase with the reference's own torch.profiler.record_function names --
 * "SAM3Image._encode_prompt", "SAM3Image._run_encoder", "SAM3Image._run_decoder", "SAM3Image._run_segmentation_heads"
 * (sam3/model/sam3_image.py:449-479) inside esam3_ground, "sam_mask_decoder" (sam3_tracker_base.py:314) inside esam3_decode -- so that
 * traces of the two implementations line up.  Every scope is forwarded to roctx (rocprofv3 --marker-trace) when libroctx64.so can be
 * loaded, and to these process-wide hooks (NULL = none): the Python layer registers callbacks that open / close a
 * torch.profiler.record_function range while a torch profiler is running.  push / pop are called on the thread inside the engine call. */
void esam3_set_scope_hooks(void (*push)(const char* name), void (*pop)(void));
/* Per-launch timing with HIP events on the launch stream (bench.py roofline leg): enable,
 * run encode/decode, then fetch a JSON report (syncs the device, clears the records). */
int esam3_profile_enable(esam3_engine* e, int on);
int esam3_profile_report(esam3_engine* e, char* json_buf, int64_t buf_size);
/* Time only the launches whose tag (= weight name or kernel tag, as listed by esam3_profile_report) equals
 * `tag`, with HIP events on the launch stream; every other launch runs un-instrumented (bench.py's
 * live roofline leg).  NULL or "" stops watching.  Clears the collected records. */
int esam3_profile_tag(esam3_engine* e, const char* tag);

/* bytes of engine workspace currently reserved, and size of one activation element */
int64_t esam3_workspace_bytes(const esam3_engine* e);
int esam3_elem_size(const esam3_engine* e);

/* ---- single-operator entry points (unit parity tests; same kernels as the engine) ---- */
/* out[M][N] = act(A[M][K] @ W[N][K]^T + bias) (+res); W/bias host fp32, A/out device engine dtype */
int esam3_op_linear(int dtype, const void* a_dev, const float* w_host, const float* bias_host,
                    const void* res_dev, void* out_dev, int64_t M, int N, int K, int act,
                    void* hip_stream);
/* bf16 only: out[M][Cout] = res + W2 act(W1 x + b1) + b2 in one launch (fused_mlp.hip: RepViT channel mixer, repvit.py:125-161;
 * TinyViT Mlp behind its norm, tiny_vit.py:196-217); w1 [Hid][Cin], w2 [Cout][Hid] host fp32; shapes 64-128-64, 128-256-128,
 * 128-512-128, 256-512-256 */
int esam3_op_fused_mlp(const void* x_dev, const float* w1_host, const float* b1_host, const float* w2_host, const float* b2_host,
                       const void* res_dev, void* out_dev, int64_t M, int Cin, int Hid, int Cout, int act, void* hip_stream);
/* bilinear resize (align_corners=False) of a ConvT-k2s2 (taps = 4, tap-major channel blocks, pixel-shuffled to 2 OH x 2 OW) or 1x1
 * (taps = 1) layer's output computed on the un-resized map: in [B][IH][IW][taps*C] -> out [B][s*OH(+2)][s*OW(+2)][C], + bias[C]
 * (host fp32 or NULL), activation, optional 1-pixel zero border (the caller zeroes it).  The commuted front end of the necks:
 * F.interpolate of model_builder.py:779-786 applied after the first layer of necks.py:42-92 instead of before it. */
/* host-only test hook: the per-axis interpolation maps of a bilinear (align_corners = False) resize in_size -> out_size as the
 * resize_shuffle row kernel receives them: first[c] / count[c] = the run of output indices whose source cell is c (c < in_size),
 * frac[o] = interpolation fraction of output o.  Same float arithmetic as ATen's area_pixel_compute_source_index
 * (the head's F.interpolate, model_builder.py:779-786).  Returns -1 outside the kernel's range. */
int esam3_resize_axis_tables(int in_size, int out_size, int* first_host, int* count_host, float* frac_host);
int esam3_op_resize_shuffle(int dtype, const void* in_dev, const float* bias_host, void* out_dev, int B, int IH, int IW, int OH, int OW,
                            int C, int taps, int act, int out_pad, void* hip_stream);
/* dense 3x3/s1/p1 or 1x1 conv, NHWC; w_host is the PyTorch [Cout][Cin][k][k] fp32 weight */
int esam3_op_conv2d(int dtype, const void* x_dev, const float* w_host, const float* bias_host,
                    const void* res_dev, void* out_dev, int B, int H, int W, int Cin, int Cout,
                    int ksize, int act, void* hip_stream);
/* 3x3/s1/p1 conv whose NHWC input already carries a 1-pixel zero border ([B][H+2][W+2][Cin]);
 * out_pad=1 writes the output inside a zero border too ([B][H+2][W+2][Cout]) */
int esam3_op_conv3x3_padded(int dtype, const void* x_padded_dev, const float* w_host,
                            const float* bias_host, void* out_dev, int B, int H, int W, int Cin,
                            int Cout, int act, int out_pad, void* hip_stream);
/* 3x3 stride-2 pad-1 conv, NHWC [B][H][W][Cin] -> [B][ceil(H/2)][ceil(W/2)][Cout]
 * (RepViT / TinyViT patch embedding, repvit.py:241-242) */
int esam3_op_conv3x3_s2(int dtype, const void* x_dev, const float* w_host, const float* bias_host,
                        void* out_dev, int B, int H, int W, int Cin, int Cout, int act, void* hip_stream);
/* TinyViT window attention, head dim 32 (tiny_vit.py:265-293,339-372): qkv [B][H][W][heads*96]
 * (q|k|v per head); positions of the zero-padded border use pad_qkv [heads*96]; bias [heads][ws*ws]
 * indexed by |dy|*ws+|dx|; out [B][H][W][heads*32]; ws in {7, 14} */
int esam3_op_window_attention(int dtype, const void* qkv_dev, const float* pad_qkv_host,
                              const float* bias_host, void* out_dev, int B, int H, int W, int heads,
                              int ws, void* hip_stream);
/* ViT-H attention (vitdet.py:466-515): qkv [B][H][W][3][heads][64]; softmax(q k^T / 8) v over ws x ws
 * windows (ws = H = W: global attention); out [B][H][W][heads*64].  bf16 runs on MFMA when ws*ws % 64 == 0.
 * With a RoPE table [ws*ws][32][2] q and k are rotated first (on the fly on MFMA; in place in qkv otherwise) */
int esam3_op_attn_window(int dtype, void* qkv_dev, const float* rope_cos_sin_host /* or NULL */, void* out_dev,
                         int B, int H, int W, int ws, int heads, void* hip_stream);
/* nn.MultiheadAttention core after the input projections, heads x 32 (the PCS encoder / decoder attentions,
 * sam3/model/encoder.py:60-130, decoder.py:120-200): q [B*Nq][heads*32], k / v [B*Nk][heads*32] in one allocation
 * (v within 2^31 elements of k), key_mask [B][Nk] (1 = ignore) or NULL, separable box-relative bias
 * bias_y [B][heads][Nq][Hk] + bias_x [B][heads][Nq][Wk] fp32 or NULL (decoder.py:333-415; query rows < bias_q0 get
 * none).  path 0: fp32-accumulate VALU kernel (any dtype); 1: bf16 MFMA flash kernel (no mask / bias, Nk % 64 == 0,
 * Nq >= 64); 2: bf16 MFMA kernel for few queries (keys split over the waves of a block).  All device pointers. */
int esam3_op_mha(int dtype, int path, const void* q_dev, const void* k_dev, const void* v_dev, void* out_dev, int B, int Nq,
                 int Nk, int heads, const uint8_t* key_mask_dev, const float* bias_y_dev, const float* bias_x_dev, int Hk,
                 int Wk, int bias_q0, void* hip_stream);
/* 2-D axial RoPE in place on the q and k parts of qkv (vitdet.py:41-90): cos_sin_host [ws*ws][32][2] */
int esam3_op_vit_rope(int dtype, void* qkv_dev, const float* cos_sin_host, int64_t rows, int H, int W, int ws,
                      int heads, void* hip_stream);
/* timm SqueezeExcite in place on x [B][HW][C] (repvit.py:136,150): w1 [R][C], w2 [C][R] host fp32 */
int esam3_op_squeeze_excite(int dtype, void* x_dev, const float* w1_host, const float* b1_host,
                            const float* w2_host, const float* b2_host, int B, int HW, int C, int R,
                            void* hip_stream);
/* the composed "up-conv" of neck level 0 (necks.py:74-98: dconv_2x2_1 -> conv_1x1 -> conv_3x3 [-> conv_s0 of mask_decoder.py]), bf16:
 * ConvTranspose2d k2 s2 (wt_host [Cin][Cmid][2][2] + bt_host [Cmid], the 1x1 already folded in by the caller) followed by a 3x3 / pad 1
 * conv (w3_host [Cout][Cmid][3][3] + b3_host [Cout]) as four 2x2 convs on the ConvT's INPUT, one per parity class of the output pixel;
 * the same host composition, packing and kernels as the engine's level-0 launches.  xpad_dev: [B][H+2][W+2][Cin] with a zero border,
 * out_dev: [B][2H][2W][Cout].  narrow = 0: gemm256p's up-conv gather (the dominant launch; Cout % 256 == 0, Cin % 64 == 0, H, W % 16 == 0),
 * narrow = 1: upconv_narrow_kernel (Cout == 32: the SAM2-side level 0). */
int esam3_op_upconv(const void* xpad_dev, const float* wt_host, const float* bt_host, const float* w3_host, const float* b3_host,
                    void* out_dev, int B, int H, int W, int Cin, int Cmid, int Cout, int narrow, void* hip_stream);
/* ConvTranspose2d k2 s2, NHWC; w_host is the PyTorch [Cin][Cout][2][2] fp32 weight */
int esam3_op_conv_transpose2x2(int dtype, const void* x_dev, const float* w_host,
                               const float* bias_host, const void* res_dev, void* out_dev, int B,
                               int H, int W, int Cin, int Cout, int act, int res_after_act,
                               void* hip_stream);
/* fused MBConv (1x1 expand + Hardswish -> dw3x3 stride 1|2 + Hardswish -> 1x1 project [+ x]), NHWC;
 * w1 [Cmid][Cin], wd [Cmid][1][3][3], w2 [Cout][Cmid] host fp32 with BatchNorm already folded */
int esam3_op_mbconv_fused(int dtype, const void* x_dev, const float* w1_host, const float* b1_host,
                          const float* wd_host, const float* bd_host, const float* w2_host,
                          const float* b2_host, void* out_dev, int B, int H, int W, int Cin, int Cmid,
                          int Cout, int stride, int residual, void* hip_stream);
/* round-4 fused MBConv on the bf16 engine (csrc/evit_fused.hip: depthwise phase on the matrix cores; EfficientViT-B0/B1 shapes
 * up to 256 channels, the local modules of the EfficientViTBlocks included); same host weight layout as esam3_op_mbconv_fused;
 * x_dev / out_dev bf16 NHWC.  Reference: backbones/efficientvit/nn/ops.py:315-367,740-770.
 * residual: bit 0 = identity shortcut (stride 1, Cin == Cout); 3 = TinyViT MBConv (backbones/tiny_vit.py:73-108: GELU after conv1, conv2 and
 * the shortcut add; 64 -> 256 -> 64); 4 = TinyViT PatchMerging (tiny_vit.py:128-154: conv1 -> GELU -> depthwise 3x3 stride 2 -> GELU -> conv3,
 * Cmid == Cout, no shortcut, no closing activation; 64 -> 128 and 128 -> 256) */
int esam3_op_mbconv3(const void* x_dev, const float* w1_host, const float* b1_host, const float* wd_host,
                     const float* bd_host, const float* w2_host, const float* b2_host, void* out_dev, int B, int H, int W,
                     int Cin, int Cmid, int Cout, int stride, int residual, void* hip_stream);
/* fused LiteMLA context module of an EfficientViTBlock on the bf16 engine (csrc/evit_fused.hip): out = x + BN(proj(relu linear
 * attention over [qkv ; aggreg(qkv)])), dim 16, C = 128 | 256.  wqkv [3C][C], wdw [3C][1][5][5], wgrp [3C][16] (groups of 16),
 * wproj [C][2C] with BatchNorm folded, bproj [C]: host fp32; x_dev / out_dev bf16 NHWC.
 * Reference: backbones/efficientvit/nn/ops.py:521-671 (LiteMLA), :740-770 (ResidualBlock) */
int esam3_op_lite_mla_block(const void* x_dev, const float* wqkv_host, const float* wdw_host, const float* wgrp_host,
                            const float* wproj_host, const float* bproj_host, void* out_dev, int B, int H, int W, int C,
                            void* hip_stream);
/* out[r] = x[r] W^T + bias + table[r mod P] for 256 -> 256 channels on the bf16 engine with the weights resident in LDS
 * (csrc/decoder_fused.hip: the merged k | v projection of the image tokens, sam/transformer.py:165-170,226-253).  x_dev / out_dev bf16
 * [rows][256], rows % 32 == 0; w [256][256], bias [256] or NULL, table [P][256] or NULL (P % 32 == 0): host fp32 */
int esam3_op_rowlin256(const void* x_dev, const float* w_host, const float* bias_host, const float* table_host, int P, void* out_dev,
                       int64_t rows, void* hip_stream);
/* "image attends to the tokens" block of the two-way transformer on the bf16 engine (csrc/decoder_fused.hip), one kernel:
 * q = (x + pe) Wq^T + bq, 8 heads x 16 attention over the T <= 16 prompt tokens (k / v already projected), out_proj, + x, LayerNorm.
 * x_dev / out_dev bf16 [Bp][P][256] (P % 16 == 0); wq [128][256], bq [128], peq [P][128] (= pe Wq^T), wo [256][128], bo [256],
 * gamma / beta [256], tk / tv [Bp][T][128]: host fp32.  Reference: sam3/sam/transformer.py:177-182 (cross_attn_image_to_token + norm4),
 * :185-253 (Attention) */
int esam3_op_i2t_block(const void* x_dev, const float* wq_host, const float* bq_host, const float* peq_host, const float* wo_host,
                       const float* bo_host, const float* gamma_host, const float* beta_host, const float* tk_host,
                       const float* tv_host, void* out_dev, int Bp, int P, int T, void* hip_stream);
/* depthwise k x k (3|5), stride 1|2; w_host PyTorch [C][1][k][k] */
int esam3_op_dwconv(int dtype, const void* x_dev, const float* w_host, const float* bias_host,
                    void* out_dev, int B, int H, int W, int C, int ksize, int stride, int act,
                    void* hip_stream);
int esam3_op_stem(int dtype, const float* img_nchw_dev, const float* w_host /*[Cout][3][3][3]*/,
                  const float* bias_host, void* out_dev, int B, int H, int W, int Cout, int act,
                  void* hip_stream);
/* EfficientViT input stem as ONE kernel: conv 3 -> 16 k3 s2 p1 (+BN folded) + Hardswish, then ResidualBlock(DSConv): depthwise 3x3 +
 * Hardswish, pointwise 16 -> 16, + identity.  img NCHW fp32 on the device; w0 [16][3][3][3], wd [16][1][3][3], wp [16][16] host fp32;
 * out NHWC [B][ceil(H/2)][ceil(W/2)][16].  variant 0 = the engine's path (bf16: persistent workgroups), 1 = the one-tile-per-workgroup
 * kernel of round 2 (kept for the bit-exactness A/B of tests/test_ops_gpu.py).
 * Reference: sam3/backbones/efficientvit/backbone.py:48-70 (input_stem), nn/ops.py:285-306 (DSConv), :740-770 (ResidualBlock) */
int esam3_op_stem_dsconv(int dtype, const float* img_nchw_dev, const float* w0_host, const float* b0_host, const float* wd_host,
                         const float* bd_host, const float* wp_host, const float* bp_host, void* out_dev, int B, int H, int W,
                         int variant, void* hip_stream);
/* LiteMLA linear attention on a [B][N][2*3*heads*dim] multi-scale qkv tensor */
int esam3_op_lite_mla(int dtype, const void* ms_dev, void* out_dev, int B, int N, int groups,
                      int dim, void* hip_stream);
int esam3_op_grouped_pw(int dtype, const void* x_dev, const float* w_host /*[C][gs]*/,
                        void* out_dev, int64_t rows, int C, int gs, void* hip_stream);
int esam3_op_resize_bilinear(int dtype, const void* x_dev, void* out_dev, int B, int IH, int IW,
                             int OH, int OW, int C, void* hip_stream);
int esam3_op_layernorm(int dtype, const void* x_dev, const void* res_dev, const float* gamma_host,
                       const float* beta_host, void* out_dev, int64_t rows, int C, float eps,
                       int act, void* hip_stream);
/* few_keys: 0 generic / LDS-tiled token -> image kernel, 1 image -> token kernel (<= 64 keys), 2 token -> image attention on the
 * matrix cores (bf16, 8 heads x 16, Nq <= 16; csrc/decoder_fused.hip), 3 the same with k_dev holding [k | v] rows of 2 x heads x
 * head_dim (the merged projection the engine writes; v_dev ignored) */
int esam3_op_attention(int dtype, const void* q_dev, const void* k_dev, const void* v_dev,
                       void* out_dev, int B, int Nq, int Nk, int heads, int head_dim,
                       int few_keys, void* hip_stream);
int esam3_op_fill_holes(const float* in_dev, float* out_dev, int n, int H, int W, float thr,
                        float max_area, void* hip_stream);
int esam3_op_upsample_masks(const float* in_dev, float* out_f32_dev, uint8_t* out_u8_dev, int n,
                            int IH, int IW, int OH, int OW, float thr, void* hip_stream);
int esam3_op_cast(int dtype, int to_f32, const void* in_dev, void* out_dev, int64_t n,
                  void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* ESAM3_H */
