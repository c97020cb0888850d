/*
 * tef.h — C ABI of libtef_hip.so, the MI355X (gfx950) implementation of the
 * tudelft/taming_event_flow training hot path.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; each entry point below
 * names the reference code it replaces (path:line under the reference tree).  The Python
 * host modules in taming_event_flow_amd/ (loss/flow.py, dataloader/encodings.py,
 * models/...) bind these with ctypes and keep the reference's module API.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless the name ends in _host;
 *   - the caller owns every buffer, including workspaces (size queries are provided);
 *     the library never allocates or frees device memory and keeps no mutable global state
 *     (except the opt-in profiling accumulators below);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no implicit sync;
 *   - return value: 0 = ok, TEF_ERR_* < 0 otherwise (never throws across the ABI);
 *     tef_last_error() returns a static description of the last failure of the calling thread.
 */
#ifndef TEF_H
#define TEF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEF_VERSION 1
#define TEF_MAX_PASSES 64     /* passes_loss upper bound */
#define TEF_MAX_SCALES 6      /* scales_loss upper bound */

#define TEF_ERR_INVALID   (-1)   /* bad argument / unsupported configuration */
#define TEF_ERR_WORKSPACE (-2)   /* workspace too small */
#define TEF_ERR_LAUNCH    (-3)   /* HIP launch failure (see tef_last_error) */

#define TEF_KIND_ITERATIVE 0     /* loss/flow.py:415 Iterative */
#define TEF_KIND_LINEAR    1     /* loss/flow.py:216 Linear    */

int tef_version(void);
const char *tef_last_error(void);

/* Opt-in per-kernel timing for bench.py's roofline line: while enabled, HIP events are recorded on the
 * launch stream around every kernel (the only mutable process state the library keeps).
 * tef_profile_enable(1) resets the accumulators; tef_profile_collect() synchronises the recorded events and
 * accumulates per-slot totals; slots are named (tef_profile_name) after the kernels in DESIGN.md. */
int tef_profile_enable(int on);
int tef_profile_pause(int paused);      /* stop / resume recording without clearing what was recorded (sampled timing) */
int tef_profile_collect(void);
/* Per-layer attribution of the network passes (round 6): while profiling is on, tef_net_pass_forward / _backward /
 * tef_net_window_wgrads bracket every layer and direction ("enc1.gru.ur fwd", "dec2 dgrad", "res0.conv1 wgrad" ...) with an
 * event pair on the launch stream — meaningful when the window runs on ONE stream.  After tef_profile_collect: lines
 * "label,scopes,ms" into buf (NUL-terminated, truncated to nbytes); returns the full length. */
long tef_profile_layers(char *buf, size_t nbytes);
int tef_profile_slots(void);
const char *tef_profile_name(int slot);
double tef_profile_ms(int slot);
long tef_profile_calls(int slot);

/* Event list of one loss window in structure-of-arrays form, [B][cap] per array.
 * Slots of pass t occupy [off[t], off[t+1]) of every sample's row (see tef_loss_cfg); off[t] are multiples of 64 (a
 * wavefront of the chain kernels works on 64 slots of one pass): tef_pack_events fills the slots up to the next multiple with empty events.
 * ts already carries the "+ pass index" shift of loss/flow.py:457-458 (tef_pack_events adds it).
 * mp/mn = polarity mask columns (pos, neg) of dataloader/base.py:265-278.
 * bin[cap] = pass index of each slot (shared by all samples). */
typedef struct tef_events {
    const float *ts, *y, *x, *mp, *mn;
    const uint8_t *bin;
    const int *cls;   /* [B][TEF_MAX_PASSES][3]: per sample and pass, where the pos-only (mask exactly (1,0)) /
                         neg-only ((0,1)) / general (both polarities set, or a mask value other than 0 or 1) runs of
                         the pass end (slots relative to the pass start); written by tef_pack_events */
    int cap;
} tef_events;

/* One loss window.  Mirrors what BaseEventWarping/Iterative/Linear.__init__ read from the
 * config (loss/flow.py:25-28, 42-44, 434-441) plus the per-pass event counts. */
typedef struct tef_loss_cfg {
    int kind;                 /* TEF_KIND_* */
    int B, H, W;              /* loader.batch_size, loader.resolution */
    int P;                    /* data.passes_loss (= number of update() calls in the window) */
    int F;                    /* number of flow heads per pass (len(flow_list)) */
    int S;                    /* data.scales_loss */
    int mode_div;             /* Iterative: 1 = iterative_mode "one", 2 = "two" */
    int M, Md;                /* total grad / detached slots per sample */
    int off[TEF_MAX_PASSES + 1];    /* grad slot offsets per pass */
    int doff[TEF_MAX_PASSES + 1];   /* detached slot offsets per pass */
    int loss_scaling;               /* BaseEventWarping(loss_scaling=...), loss/flow.py:124-127: 1 = divide every image's sum
                                       by its number of active pixels (the default), 0 = plain sum */
    int border_compensation;        /* BaseEventWarping(border_compensation=...): 1 (the only value the reference's Linear /
                                       Iterative constructors produce) = an event enters the images of a window only if it
                                       stays inside the frame at EVERY reference time of the window (loss/flow.py:671-681,
                                       Linear: at both window ends, :341-343);
                                       0 = Iterative: at each reference time on its own (:691-693, :709-711); Linear: nothing
                                       is purged, corners outside the frame drop out one by one (:324-328) */
} tef_loss_cfg;

/* AoS -> SoA packing of one pass, replaces the bookkeeping of Iterative.update / Linear.update
 * (loss/flow.py:443-476, 233-288): adds `ts_shift` to ev[:, :, 0] IN PLACE (reference side effect,
 * :457-458) and appends the pass at slot `slot0` of the SoA arrays.  If ts_override (a DEVICE pointer to one float) is
 * not NULL the stored timestamp is that value instead (round_ts, :461-463: the caller computes min + 0.5 of the shifted
 * list on the device, no host round trip).  ev [B,N,4] (ts,y,x,p), pm [B,N,2].  slot0 must be a multiple of 64 and
 * cap >= slot0 + N rounded up to 64: the next pass starts there.
 * The events of the pass are stored sorted by (polarity class, 16x8 pixel tile of the H x W frame, pixel row): the loss
 * is a sum over events, so the order inside a pass is free, and coherent wavefronts halve the cost of the lookups. */
int tef_pack_events(float *ev, const float *pm, int B, int N, float ts_shift, const float *ts_override, int pass_idx,
                    int slot0, int cap, int H, int W, float *ts, float *y, float *x, float *mp, float *mn,
                    uint8_t *bin, int *cls, void *stream);

/* Flow map of one head of one pass, replaces BaseEventWarping.update_base (loss/flow.py:46-66): copies the
 * model's [B,2,H,W] tensor (channel 0 = x, 1 = y; element strides stride_b / stride_c, rows dense) into
 *   planar [B][2][H*W]            -> slot [t][i] of the planar window buffer  [P][F][B][2][H][W]  (smoothing terms)
 *   yx     [B][H*W] (flow_y, flow_x) pairs -> slot [t][i] of the interleaved buffer [P][F][B][H][W][2]  (lookups) */
int tef_pack_flow(const float *flow, long stride_b, long stride_c, int B, int H, int W, float *planar, float *yx,
                  void *stream);

/* The same for all F heads of one pass in one launch: flows / stride_b / stride_c are HOST arrays of F entries; planar and
 * yx point at slot [t][0] of the window buffers (head i lands at slot [t][i]). */
int tef_pack_flows(const float *const *flows, const long *stride_b, const long *stride_c, int F, int B, int H, int W,
                   float *planar, float *yx, void *stream);

/* One pass of Iterative.update / Linear.update (loss/flow.py:443-476 / :233-288) as ONE call: the F flow maps of the
 * pass (tef_pack_flows) and its two event lists (tef_pack_events, in-place time shift by pass_idx included) — three
 * launches from one host call instead of three calls with their argument marshalling.  grad / det: the SoA stores of the
 * window (written at slots slot0 / dslot0 on); planar / yx: this pass's slices of the window's flow buffers. */
int tef_update_pass(const float *const *flows, const long *stride_b, const long *stride_c, int F, int B, int H, int W,
                    float *planar, float *yx, float *ev, const float *pm, int N, const float *ts_override, float *dev,
                    const float *dpm, int Nd, const float *dts_override, int pass_idx, int slot0, int dslot0,
                    const tef_events *grad, const tef_events *det, void *stream);

/* One pass of a window for tef_update_window: the per-pass arguments of tef_update_pass as a record (HOST memory; flows /
 * stride_b / stride_c are host arrays of F entries). */
typedef struct tef_update_desc {
    const float *const *flows;
    const long *stride_b, *stride_c;
    float *ev;
    const float *pm;
    const float *ts_override;
    float *dev;
    const float *dpm;
    const float *dts_override;
    int N, Nd, pass_idx, slot0, dslot0;
} tef_update_desc;

/* All the passes of a window (loss/flow.py:443-476 called passes_loss times) in ONE launch per ~20 passes: for a caller
 * that holds the whole window — a staged window, a loss-only caller, an update() whose work is deferred to the evaluation
 * (taming_event_flow_amd.loss.flow: `defer_update`).  Same effect as tef_update_pass per record, the in-place time shift
 * of the callers' event lists included.  planar / yx: slot [0][0] of the window's flow buffers (pass t lands at [t]). */
int tef_update_window(const tef_update_desc *passes, int npass, int F, int B, int H, int W, float *planar, float *yx,
                      const tef_events *grad, const tef_events *det, void *stream);

/* Workspace (bytes) needed by tef_loss_forward + tef_loss_backward for this window. */
size_t tef_loss_workspace_bytes(const tef_loss_cfg *cfg);

/* Contrast-maximisation loss forward: Iterative.forward (loss/flow.py:588-736) or Linear.forward (:306-402),
 * without the optional smoothing terms (see tef_smoothing_*).
 *   flows_yx  [P][F][B][H][W][2]  interleaved (flow_y, flow_x), see tef_pack_flow (already multiplied by
 *             flow_scaling by the caller, train_flow.py:107-108)
 *   grad / det: event lists with / without gradient (det may have Md = 0)
 *   loss_out: one float.  The workspace keeps what tef_loss_backward needs. */
int tef_loss_forward(const tef_loss_cfg *cfg, const float *flows_yx, const tef_events *grad, const tef_events *det,
                     void *workspace, size_t workspace_bytes, float *loss_out, void *stream);

/* d loss / d flows as PLANAR [P][F][B][2][H][W] (channel 0 = x, 1 = y: the layout of the model's tensors);
 * grad_out = upstream scalar gradient (device pointer).
 * Must follow tef_loss_forward on the same workspace.  dflows is fully overwritten.  A call that returned an error leaves
 * the workspace's work lists half-consumed: run tef_loss_forward again before retrying (a second backward on an intact
 * workspace is fine). */
int tef_loss_backward(const tef_loss_cfg *cfg, const float *flows_yx, const tef_events *grad, const tef_events *det,
                      void *workspace, size_t workspace_bytes, const float *grad_out, float *dflows, void *stream);

/* Optional Charbonnier priors of loss/flow.py:170-209 (spatial) and :131-168 (temporal) on the flow maps of
 * the window (flows = the PLANAR buffer [P][F][B][2][H][W]).  weight < 0 disables a term.  loss_out += term (accumulates onto the CM loss);
 * backward ADDS into dflows.  scratch: tef_smoothing_scratch_bytes(). */
size_t tef_smoothing_scratch_bytes(const tef_loss_cfg *cfg);
int tef_smoothing_forward(const tef_loss_cfg *cfg, const float *flows, float spat_weight, float temp_weight,
                          void *scratch, float *loss_out, void *stream);
int tef_smoothing_backward(const tef_loss_cfg *cfg, const float *flows, float spat_weight, float temp_weight,
                           void *scratch, const float *grad_out, float *dflows, void *stream);

/* Input representations of dataloader/encodings.py, for one sample (B = 1, the reference's call shape) or a
 * zero-padded batch.  Element e of sample b of each array sits at base[b * batch_stride + e * elem_stride]
 * (SoA: elem_stride 1; the collated AoS list [B,N,4] = (ts,y,x,p): elem_stride 4 and xs = ev + 2, ys = ev + 1, ...).
 *   TEF_ENCODE_IMAGE     events_to_image    encodings.py:8-29    out [B][1][H][W]        += p
 *   TEF_ENCODE_CHANNELS  events_to_channels encodings.py:59-81   out [B][2][H][W]        per-polarity counts
 *   TEF_ENCODE_VOXEL     events_to_voxel    encodings.py:32-56   out [B][channels][H][W] temporal bilinear, ts in [0,1]
 * out is fully overwritten. */
#define TEF_ENCODE_IMAGE 0
#define TEF_ENCODE_CHANNELS 1
#define TEF_ENCODE_VOXEL 2
int tef_encode_events(const float *xs, const float *ys, const float *ts, const float *ps, int B, long batch_stride,
                      int elem_stride, int N, int mode, int channels, int H, int W, float *out, void *stream);
/* The same representations from the two collated lists of one batch, event_list [B][N][4] and d_event_list [B][Nd][4]
 * (either may be empty): the loader encodes all events of the window before it splits them (h5.py:371-385). */
int tef_encode_event_lists(const float *event_list, int N, const float *d_event_list, int Nd, int B, int mode,
                           int channels, int H, int W, float *out, void *stream);

/* ---- loader stage: format, augment, split and collate the raw events of a whole batch ------------------------------
 * Raw events of sample b are elements [offsets[b], offsets[b+1]) of xs, ys, ts, ps (device, fp32; ps in {0,1}, ts
 * increasing); offsets (host, offsets[0] = 0), flags (host, TEF_AUG_* bits per sample or NULL).  Per sample, as one
 * __getitem__ of the reference does (dataloader/h5.py:340-345,349-366,413-416; base.py:153-177,192-222,252-278,348-377):
 * <= 10 events -> none; p = 2 ps - 1; ts = (ts - ts[0]) / (ts[-1] - ts[0]); flips; event (ts,y,x,p), mask (p>0, p<0);
 * with max_grad > 0 and more than max_grad events, event sampled[b][j] (device int32 [B][max_grad], the caller's
 * multinomial draw, distinct) becomes gradient event j and the others form the detached list in stream order.  Then
 * custom_collate (base.py:392-434): lists zero-padded to N / Nd (>= the values tef_collate_counts returns).
 * Outputs event_list [B][N][4], pol_mask [B][N][2], d_event_list [B][Nd][4], d_pol_mask [B][Nd][2], fully written. */
#define TEF_MAX_BATCH 64
#define TEF_AUG_HORIZONTAL 1   /* x -> W-1-x   base.py:206-210 */
#define TEF_AUG_VERTICAL   2   /* y -> H-1-y   base.py:212-216 */
#define TEF_AUG_POLARITY   4   /* p -> -p      base.py:218-220 */
int tef_collate_counts(const int *offsets, int B, int max_grad, int *n_grad, int *n_detached);
size_t tef_collate_workspace_bytes(const int *offsets, int B);
int tef_collate_events(const float *xs, const float *ys, const float *ts, const float *ps, const int *offsets,
                       const int *sampled, const int *flags, int B, int max_grad, int N, int Nd, int H, int W,
                       void *workspace, size_t workspace_bytes, float *event_list, float *pol_mask,
                       float *d_event_list, float *d_pol_mask, void *stream);

/* ---- RecEVFlowNet convolutions (models/submodules.py) -------------------------------------------------------
 * One 3x3 / 1x1 convolution with padding ksize/2, stride 1 or 2, bias and a fused activation, on NCHW fp32 tensors.
 * The input is the channel concatenation of up to two tensors, x0 [B,C0,H,W] and x1 [B,C1,H,W] (C1 may be 0), the
 * second optionally multiplied element-wise by gate1 [B,C1,H,W]: this is exactly what ConvGRU feeds its gates,
 * torch.cat([input_, prev_state]) and torch.cat([input_, prev_state * reset]) (submodules.py:146,149), without
 * materialising the concatenation.  weight is the nn.Conv2d parameter [N, C0+C1, k, k], bias [N] (may be NULL).
 * Replaces ConvLayer.forward :53-62, the three gate convolutions of ConvGRU.forward :134-152, ResidualBlock
 * conv1/conv2 :207-227 and the conv of UpsampleConvLayer.forward :263-273, and autograd through them.
 * Arithmetic: fp32 operands, fp32 accumulation on v_mfma_f32_32x32x2_f32. */
#define TEF_ACT_NONE 0
#define TEF_ACT_RELU 1
#define TEF_ACT_TANH 2
#define TEF_ACT_SIGMOID 3
typedef struct tef_conv_desc {
    int B, C0, C1, H, W;   /* input */
    int N;                 /* output channels */
    int ksize, stride;     /* 1 or 3; 1 or 2 */
    int act;               /* TEF_ACT_* applied to conv + bias */
} tef_conv_desc;

size_t tef_conv_workspace_bytes(const tef_conv_desc *d);
/* GEMM operands of the weight: wp = [N][Kp], k = (ci, ky, kx) padded to 16 (forward) and w2 = [C0+C1][K2p],
 * k' = (n, ky, kx) padded to 16 (input gradient).  Sizes in floats via tef_conv_packed_weight_floats (returns their
 * sum).  Pack once per optimiser step, not per call.  A row-concatenation of several parameters (ConvGRU
 * update|reset gates sharing one GEMM) is packed part by part: `rows` rows of `weight` land at row `row0`. */
size_t tef_conv_packed_weight_floats(const tef_conv_desc *d, size_t *wp_floats, size_t *w2_floats);
int tef_conv_pack_weight(const tef_conv_desc *d, const float *weight, int rows, int row0, float *wp, float *w2,
                         void *stream);
/* The same for any number of weight parts (of any layers) in ONE launch: after an optimiser step a network re-packs every
 * layer, and ~50 small launches one after the other kept the chip nearly empty for ~1 ms in front of the first convolution
 * (round 6).  jobs: HOST array; a job = one tef_conv_pack_weight call's arguments. */
typedef struct {
    tef_conv_desc desc;
    const float *weight;
    int rows, row0;
    float *wp, *w2;
} tef_pack_job;
int tef_conv_pack_weights(const tef_pack_job *jobs, int njobs, void *stream);
/* out [B,N,Ho,Wo] = act(conv(cat[x0, x1 * gate1], weight) + bias), weight given as its packed form wp */
int tef_conv_forward(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1, const float *wp,
                     const float *bias, float *out, void *workspace, size_t workspace_bytes, void *stream);
/* Given dout = d loss / d out (and out itself when act != NONE):
 *   dx0 [B,C0,H,W], dx1 [B,C1,H,W] = gradient w.r.t. the concatenated conv input (dx1 is w.r.t. x1 * gate1; either may
 *   together be NULL to skip; needs the packed weight w2), overwritten;  dweight [N,C0+C1,k,k] and dbias [N] are ACCUMULATED into (+=), NULL to skip. */
/* The same with the output channels delivered as two tensors, out [B,out_split,Ho,Wo] and out2 [B,N-out_split,Ho,Wo]:
 * the update | reset gates of ConvGRU come out of one GEMM as the two tensors the cell uses (submodules.py:146-148). */
int tef_conv_forward_split(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                           const float *wp, const float *bias, float *out, float *out2, int out_split, void *workspace,
                           size_t workspace_bytes, void *stream);
/* The same with the ConvGRU state update folded into the epilogue (one output tensor, out_split = N): beside
 * out = act(...) the kernel stores bl_out = bl_h * (1 - bl_u) + out * bl_u, all [B,N,Ho,Wo] (submodules.py:150: the out
 * gate's convolution delivers the new state, no separate blend pass).  bl_* all NULL = tef_conv_forward_split. */
int tef_conv_forward_blend(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1, const float *wp,
                           const float *bias, float *out, float *out2, int out_split, const float *bl_h, const float *bl_u,
                           float *bl_out, void *workspace, size_t workspace_bytes, void *stream);
int tef_conv_backward(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                      const float *w2, const float *out, const float *dout, float *dx0, float *dx1, float *dweight,
                      float *dbias, void *workspace, size_t workspace_bytes, void *stream);
/* The same for a convolution whose output channels are split: (out, dout) cover channels [0, io_split), (out2, dout2)
 * the rest (io_split = N, out2 = dout2 = NULL for one tensor); and with two accumulation targets: output channels
 * [0, split_rows) add into dweight / dbias, channels [split_rows, N) into dweight2 / dbias2 — the gradients of two
 * row-concatenated parameters (ConvGRU update | reset gates, submodules.py:122-123) go straight into their own .grad
 * buffers, no temporary, no extra add. */
int tef_conv_backward_split(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                            const float *w2, const float *out, const float *out2, const float *dout, const float *dout2,
                            int io_split, float *dx0, float *dx1, float *dweight, float *dweight2, float *dbias,
                            float *dbias2, int split_rows, void *workspace, size_t workspace_bytes, void *stream);

/* Deferred weight gradient (BPTT windows: one long reduction over the passes instead of one short reduction per pass).
 * tef_conv_backward_keep is tef_conv_backward_split with g = dY * act'(out) written to g_keep ([B][N][Ho][Wo], caller
 * owned) instead of the workspace whenever the call forms it (an activation, or outputs in two tensors; otherwise g is
 * dout itself); pass dweight = NULL there and hand the kept g and the layer inputs of up to TEF_CONV_MAX_PARTS calls to
 * tef_conv_wgrad_parts, which adds sum over parts of g_p (x) gather(x_p) into dweight (/ dweight2 beyond split_rows).
 * Only layers on the LDS-halo weight-gradient kernels (tef_conv_wgrad_parts_supported). */
#define TEF_CONV_MAX_PARTS 16
int tef_conv_backward_keep(const tef_conv_desc *d, const float *x0, const float *x1, const float *gate1,
                           const float *w2, const float *out, const float *out2, const float *dout, const float *dout2,
                           int io_split, float *dx0, float *dx1, float *dweight, float *dweight2, float *dbias,
                           float *dbias2, int split_rows, float *g_keep, void *workspace, size_t workspace_bytes,
                           void *stream);
int tef_conv_wgrad_parts_supported(const tef_conv_desc *d);
int tef_conv_wgrad_parts(const tef_conv_desc *d, int nparts, const float *const *g, const float *const *x0,
                         const float *const *x1, const float *const *gate1, float *dweight, float *dweight2,
                         int split_rows, void *stream);
/* ConvGRU state update new_state = prev * (1 - update) + out_inputs * update (submodules.py:150) and its backward
 * (dh = direct path only; the paths through the gates go through tef_conv_backward). n = element count. */
int tef_gru_blend(const float *h, const float *u, const float *o, size_t n, float *out, void *stream);
int tef_gru_blend_backward(const float *dhn, const float *h, const float *u, const float *o, size_t n, float *dh,
                           float *du, float *dout, void *stream);

/* Bilinear up-sampling by integer factors, align_corners=False, times a scalar: F.interpolate as used by
 * UpsampleConvLayer.forward (models/submodules.py:264) and RecEVFlowNet.forward (models/model.py:77-81).
 * x [planes][H][W] -> y [planes][H*scale_h][W*scale_w]; backward is the exact adjoint (dy -> dx). */
int tef_upsample_bilinear(const float *x, int planes, int H, int W, int scale_h, int scale_w, float mul, float *y,
                          void *stream);
int tef_upsample_bilinear_backward(const float *dy, int planes, int H, int W, int scale_h, int scale_w, float mul,
                                   float *dx, void *stream);

/* The same with a second addend and a crop: y [planes][H*scale_h - crop_top][W*scale_w - crop_left] =
 * mul * upsample(x + x2)[crop_top:, crop_left:] (x2 may be NULL).  The sum is the decoder's skip connection
 * (models/arch.py:236 skip_fn "sum" followed by UpsampleConvLayer's interpolate); the crop undoes the top / left padding of
 * RecEVFlowNet.forward (models/model_util.py:52-65, models/model.py:83).  Backward: dy is the cropped tensor. */
int tef_upsample_bilinear_crop(const float *x, const float *x2, int planes, int H, int W, int scale_h, int scale_w,
                               float mul, int crop_top, int crop_left, float *y, void *stream);
int tef_upsample_bilinear_crop_backward(const float *dy, int planes, int H, int W, int scale_h, int scale_w, float mul,
                                        int crop_top, int crop_left, float *dx, void *stream);

/* The two x2 up-samplings of a decoder level of RecEVFlowNet (models/arch.py:236-238: features + encoder skip, and the
 * previous prediction; align_corners=False, models/submodules.py:264) and their adjoints as ONE launch each: tensors a and b
 * share H x W, b has no second addend. */
int tef_upsample2x_pair(const float *xa, const float *xa2, int planes_a, float *ya, const float *xb, int planes_b, float *yb,
                        int H, int W, void *stream);
int tef_upsample2x_pair_backward(const float *dya, int planes_a, float *dxa, const float *dyb, int planes_b, float *dxb, int H,
                                 int W, void *stream);

/* ---- fused ConvGRU cell: ConvGRU.forward, models/submodules.py:134-152, and its backward -------------------------------
 * x (the cell input) and h (previous state, zeros for a fresh sequence, :141-143) are [B,C,H,W]; gates are 3x3.
 * Forward: (u, r) = sigmoid(conv([x, h]; update | reset weights)), o = tanh(conv([x, h * r]; out weights)),
 * hn = h * (1 - u) + o * u.  wp_ur / wp_o are packed weights (tef_conv_pack_weight with N = 2C rows update-then-reset,
 * and N = C), bias_ur = [2C] update-then-reset.  u, r, o are kept by the caller for the backward. */
typedef struct tef_gru_desc {
    int B, C, H, W;
} tef_gru_desc;
size_t tef_convgru_workspace_bytes(const tef_gru_desc *d);
int tef_convgru_cell_fwd(const tef_gru_desc *d, const float *x, const float *h, const float *wp_ur, const float *wp_o,
                         const float *bias_ur, const float *bias_o, float *u, float *r, float *o, float *hn,
                         void *workspace, size_t workspace_bytes, void *stream);
/* Backward.  dhn: HOST array of ndhn (1..4) device pointers whose SUM is d loss / d hn (the new state feeds the next
 * encoder, a decoder skip connection and the next pass: the addends are summed where they are consumed).
 * Outputs: dx, dh (overwritten); g_ur [B,2C,H,W] and g_o [B,C,H,W] = gradients w.r.t. the gate pre-activations, kept for
 * tef_conv_wgrad_parts when dw_* are NULL (BPTT window: one weight-gradient reduction per layer over all passes),
 * otherwise dw_u / dw_r / dw_o [C,2C,3,3] are accumulated (+=) here; db_* [C] accumulated (+=), may be NULL. */
int tef_convgru_cell_bwd(const tef_gru_desc *d, const float *x, const float *h, const float *u, const float *r,
                         const float *o, const float *const *dhn, int ndhn, const float *w2_ur, const float *w2_o,
                         float *g_ur, float *g_o, float *dx, float *dh, float *dw_u, float *dw_r, float *dw_o,
                         float *db_u, float *db_r, float *db_o, void *workspace, size_t workspace_bytes, void *stream);
/* The same with the layer that PRODUCED x folded into the last sweep (round 6): when g_x is not NULL the call writes
 * g_x [B,C,H,W] = x_act'(x) * (d loss / d x) — the pre-activation gradient of x's producer, e.g. the relu of a level's strided
 * head convolution (models/submodules.py:95-101) — and db_x [C] += its per-channel sums (may be NULL) instead of dx (which may
 * then be NULL): tef_grad_act's operations on the same values, one launch and one round trip of dx fewer. */
int tef_convgru_cell_bwd_head(const tef_gru_desc *d, const float *x, const float *h, const float *u, const float *r,
                              const float *o, const float *const *dhn, int ndhn, const float *w2_ur, const float *w2_o,
                              float *g_ur, float *g_o, float *dx, float *dh, float *dw_u, float *dw_r, float *dw_o,
                              float *db_u, float *db_r, float *db_o, int x_act, float *g_x, float *db_x, void *workspace,
                              size_t workspace_bytes, void *stream);
/* g [B,C,HW] = (sum of the ndy (1..4) tensors dy[k]) * act'(out), dbias [C] += per-channel sums of g (NULL to skip): the
 * pre-activation gradient of any conv layer whose output has several consumers (then tef_conv_backward* with
 * TEF_ACT_NONE on g).  dy: HOST array of device pointers.  out = the layer's activated output (unused for TEF_ACT_NONE). */
int tef_grad_act(const float *const *dy, int ndy, const float *out, int act, int B, int C, int HW, float *g, float *dbias,
                 void *stream);
/* Input gradient of a convolution whose input has ONE other role: it is the activated output of the layer before it
 * (round 6: the residual blocks, models/submodules.py:207-227 — conv -> relu -> conv).  Instead of d loss / d x the call
 * delivers that layer's pre-activation gradient
 *     g_out [B,C0,H,W] = act'(mask) * (d loss / d x  +  addend)        dbias [C0] += per-channel sums of g_out
 * (mask = the activated tensor x itself; addend [B,C0,H,W] = a gradient reaching x from another consumer, or NULL; dbias may
 * be NULL) — tef_grad_act's operations on the same values.  Where the convolution is split over its reduction (the deep
 * levels) this rides on the slabs' reduction launch; otherwise it costs one in-place sweep.  g [B,N,Ho,Wo]: the convolution's
 * own pre-activation gradient (already formed: d->act must be TEF_ACT_NONE); dweight as in tef_conv_backward (NULL: skip /
 * deferred).  One source (C1 = 0), not a 1x1 head. */
typedef struct {
    const float *mask;
    int act;
    const float *addend;
    float *g_out;
    float *dbias;
} tef_conv_post;
int tef_conv_backward_post(const tef_conv_desc *d, const float *x0, const float *w2, const float *g, float *dweight,
                           const tef_conv_post *post, void *workspace, size_t workspace_bytes, void *stream);
/* The tail of a decoder level, backward, in ONE launch (round 6; models/arch.py:238-240: decoder convolution -> activation ->
 * 1x1 prediction head -> activation).  `head` describes the 1x1 head (ksize 1, stride 1, C1 = 0, N <= 4, act = the head's
 * activation); dec [B,C0,H,W] is the decoder convolution's ACTIVATED output (the head's input), pred [B,N,H,W] the head's.
 *   gp      = (sum of the ndpred (1..4) tensors dpred[k]) * act'(pred)                  pre-activation gradient of the head
 *   db_pred += per-channel sums of gp;   dw_pred [N,C0] += gp x dec over pixels         (either may be NULL)
 *   gd      = dec_act'(dec) * (w2^T gp + dfeat)      dfeat: gradient reaching dec from its other consumer, or NULL
 *   db_dec  += per-channel sums of gd                                                    (may be NULL)
 * = tef_grad_act + tef_conv_backward_keep (1x1) + tef_grad_act, element for element the same operations in the same order.
 * dpred: HOST array of device pointers; w2: the head's packed input-gradient operand (tef_conv_pack_weight). */
int tef_dec_head_backward(const tef_conv_desc *head, const float *const *dpred, int ndpred, const float *pred, const float *w2,
                          const float *dec, int dec_act, const float *dfeat, float *gp, float *gd, float *db_pred, float *dw_pred,
                          float *db_dec, void *stream);
/* out = act(a + b), n elements: the residual connection of ResidualBlock (models/submodules.py:219-226). */
int tef_add_act(const float *a, const float *b, int act, size_t n, float *out, void *stream);

/* ---- RecEVFlowNet: one recurrent pass as ONE call (models/arch.py:217-242 + models/model.py:65-85) -------------------
 * The layer table of the reference's default architecture — `levels` x (strided 3x3 head convolution, ConvGRU cell),
 * `nres` residual blocks, `levels` x (bilinear x2 of (features + encoder skip) [and of the previous prediction], 3x3
 * convolution over the two sources, 1x1 prediction head), each prediction brought to the input size x 2^level x
 * flow_scale and cropped — walked in C: every launch of the pass is enqueued from a single call.  The caller owns
 *   - the packed weights / biases / gradient accumulators of every convolution (tef_net_conv; tef_conv_pack_weight),
 *   - ONE activation arena per pass ("tape", tef_net_tape_floats floats: everything the backward reads + the pass's
 *     outputs: the flows and the new states live in it at the offsets tef_net_layout reports),
 *   - ONE gradient arena per backward call (tef_net_gtape_floats floats: pre-activation gradients, the gradients w.r.t.
 *     the incoming states and the input at the reported offsets, scratch),
 *   - one workspace (tef_net_workspace_bytes) shared by all launches of a stream. */
#define TEF_NET_MAX_LEVELS 6
#define TEF_NET_MAX_RES 4
typedef struct tef_net_conv {
    const float *wp, *w2;        /* packed weight: forward / input-gradient GEMM operands (tef_conv_pack_weight) */
    const float *bias;           /* [N] or NULL (ConvGRU update|reset gates: the two biases concatenated) */
    float *dw, *dw2;             /* weight-gradient accumulators (+=); dw2: second row block (reset gate) or NULL */
    float *db, *db2;             /* bias-gradient accumulators (+=) or NULL */
    int defer;                   /* 1: leave the weight gradient to tef_net_window_wgrads (the pre-activation gradient
                                    stays in the gradient arena); layers tef_conv_wgrad_parts cannot run accumulate anyway */
} tef_net_conv;
typedef struct tef_net_plan {
    int B, H, W;                 /* the (padded) input: sides multiples of 2^levels */
    int bins, levels, nres, nout, final_act;
    int width[TEF_NET_MAX_LEVELS];      /* channels of the encoder levels */
    int dec_out[TEF_NET_MAX_LEVELS];    /* output channels of the decoders, coarse to fine */
    int crop_top, crop_left;     /* rows / columns of padding to drop from the full-resolution flows */
    float flow_scale;            /* extra factor on the flows (train_flow.py:107-108 flow_scaling) */
    tef_net_conv head[TEF_NET_MAX_LEVELS], gate_ur[TEF_NET_MAX_LEVELS], gate_o[TEF_NET_MAX_LEVELS];
    tef_net_conv res1[TEF_NET_MAX_RES], res2[TEF_NET_MAX_RES];
    tef_net_conv dec[TEF_NET_MAX_LEVELS], pred[TEF_NET_MAX_LEVELS];
    /* (round 6) the decoder half of SEVERAL passes as one batch.  Only the recurrent states cross passes (models/arch.py:
     * 225-227), so the residual blocks, decoders and heads of the P passes of a loss window are independent of each other:
     * with B = P x (samples) and hn_ext[i] = the passes' new states of level i stacked along the batch ([P*B, C_i, h_i, w_i]),
     * tef_net_pass_forward_part / _backward_part / tef_net_window_wgrads_part with TEF_NET_DECODERS run that half ONCE per
     * window — ten times fewer launches, and the deep levels' 512-pixel GEMMs (split over k 16..32 ways, 50 TFLOP/s) become
     * 5120-pixel ones (110 TFLOP/s).  dec_only = 1: the arenas hold no encoder buffers (tef_net_tape_floats etc. follow). */
    const float *hn_ext[TEF_NET_MAX_LEVELS];
    int dec_only;
    /* (round 6) 3x3 layers whose weight gradient the multi-part kernels cannot run (the deepest stride-2 head: 8 x 8 outputs)
     * used to reduce per pass, 45 us each for 1.2 GFLOP.  copy_batch = 1: the caller promises to hand tef_net_window_wgrads a
     * workspace (wgrad_ws / wgrad_ws_bytes >= tef_net_window_wgrads_workspace) — such a layer is then deferred too: the
     * passes' pre-activation gradients and inputs (a few MB) are copied side by side and reduced as ONE batch. */
    int copy_batch;
    void *wgrad_ws;
    size_t wgrad_ws_bytes;
} tef_net_plan;
size_t tef_net_tape_floats(const tef_net_plan *p);
size_t tef_net_gtape_floats(const tef_net_plan *p);
size_t tef_net_workspace_bytes(const tef_net_plan *p);
/* offsets in floats: flows [levels] (coarse to fine, [B,nout,H-crop_top,W-crop_left]) and new states [levels] inside a
 * tape; gradients w.r.t. the incoming states [levels] and w.r.t. the input ([B,bins,H,W]) inside a gradient arena */
int tef_net_layout(const tef_net_plan *p, size_t *flow_off, size_t *state_off, size_t *dstate_off, size_t *dx_off);
/* x [B,bins,H,W]; states_in: HOST array of `levels` device pointers (zeros for a fresh sequence, submodules.py:141-143) */
int tef_net_pass_forward(const tef_net_plan *p, const float *x, const float *const *states_in, float *tape,
                         void *workspace, size_t workspace_bytes, void *stream);
/* dflows / dstates: HOST arrays of `levels` device pointers, NULL entries = no gradient arrives there.  ran: bit mask of
 * the layers whose pre-activation gradient was formed (for tef_net_window_wgrads); dstate_valid [levels] / dx_valid: which
 * of the reported gradients were written (a dead chain leaves them unset). */
int tef_net_pass_backward(const tef_net_plan *p, const float *x, const float *const *states_in, const float *tape,
                          const float *const *dflows, const float *const *dstates, int want_dx, float *gtape,
                          unsigned long long *ran, int *dstate_valid, int *dx_valid, void *workspace,
                          size_t workspace_bytes, void *stream);
/* The two halves of a pass as separate calls, so that a caller can put them on two streams: the encoders of pass t + 1 do
 * not depend on the residual blocks / decoders / heads of pass t (only the recurrent states cross passes,
 * models/arch.py:225-227), so the halves of consecutive passes can fill each other's launch ramps and tails.
 * part: TEF_NET_ENCODERS (heads + ConvGRU cells: reads x and states_in, writes the new states into the tape),
 * TEF_NET_DECODERS (residual blocks, decoders, prediction heads, flows: reads the new states from the SAME tape), or both
 * (= tef_net_pass_forward / _backward).  The halves of one pass share the pass's tape and gradient arena; the caller orders
 * them (decoders after encoders going forward, encoders after decoders going backward) and gives each stream its own
 * workspace.
 * Backward, decoder half: dflows -> dstate_off[i] = offset (floats) in gtape of what this half sends to NEW state i, as ONE
 * tensor per level (-1: nothing), x / states_in / dstates / dx_valid unused (may be NULL).  The caller adds the gradients
 * arriving from the next pass (autograd does) and hands the sums to the encoder half as `dstates`:
 * encoder half: dstates = TOTAL gradient w.r.t. new state i -> dstate_off[i] = offset of d loss / d INCOMING state i (-1:
 * none), dx as in tef_net_pass_backward; dflows unused.  `ran` of the two calls are OR-ed for tef_net_window_wgrads. */
#define TEF_NET_ENCODERS 1
#define TEF_NET_DECODERS 2
int tef_net_pass_forward_part(const tef_net_plan *p, int part, const float *x, const float *const *states_in, float *tape,
                              void *workspace, size_t workspace_bytes, void *stream);
int tef_net_pass_backward_part(const tef_net_plan *p, int part, const float *x, const float *const *states_in, const float *tape,
                               const float *const *dflows, const float *const *dstates, int want_dx, float *gtape,
                               unsigned long long *ran, long long *dstate_off, int *dx_valid, void *workspace,
                               size_t workspace_bytes, void *stream);
/* tef_net_pass_backward_part with a SECOND set of gradients w.r.t. the new states (dstates2, NULL or one pointer or NULL per
 * level), added where dstates are: the batched decoder half of a loss window hands its share to the encoder halves directly
 * (train.Trainer's window mode), the gradients arriving from the next pass come through autograd as before — no sum of the
 * two as a launch of its own. */
int tef_net_pass_backward_part2(const tef_net_plan *p, int part, const float *x, const float *const *states_in, const float *tape,
                                const float *const *dflows, const float *const *dstates, const float *const *dstates2,
                                int want_dx, float *gtape, unsigned long long *ran, long long *dstate_off, int *dx_valid,
                                void *workspace, size_t workspace_bytes, void *stream);
/* The encoder half of a pass by LEVEL RANGE [lo, hi) — for a caller that pipelines the levels of consecutive passes over
 * streams: level i of pass t + 1 needs only level i - 1 of pass t + 1 and level i of pass t (models/arch.py:217-227), so the
 * lower levels of pass t + 1 can run beside the upper levels of pass t.  All ranges of a pass share its tape and gradient
 * arena; forward: ranges in ascending order (level lo reads state lo - 1 from the tape; lo = 0 reads x); backward: descending.
 * Backward: above_valid = the dx_valid the call for [hi, ...) returned (ignored when hi is the top): the gradient w.r.t.
 * state hi - 1 it left in the arena is added at level hi - 1.  *dx_valid: lo = 0: as tef_net_pass_backward; lo > 0: whether
 * such a gradient was left for the levels below.  dstates / dstates2 / dstate_off are indexed by level as in _part2. */
int tef_net_pass_forward_levels(const tef_net_plan *p, int lo, int hi, const float *x, const float *const *states_in, float *tape,
                                void *workspace, size_t workspace_bytes, void *stream);
int tef_net_pass_backward_levels(const tef_net_plan *p, int lo, int hi, int above_valid, const float *x,
                                 const float *const *states_in, const float *tape, const float *const *dstates,
                                 const float *const *dstates2, int want_dx, float *gtape, unsigned long long *ran,
                                 long long *dstate_off, int *dx_valid, void *workspace, size_t workspace_bytes, void *stream);
/* the deferred weight gradients of a BPTT window: per layer one reduction over the pixels of all npass backward calls */
int tef_net_window_wgrads(const tef_net_plan *p, int npass, const float *const *x, const float *const *const *states_in,
                          const float *const *tape, const float *const *gtape, const unsigned long long *ran,
                          void *stream);
/* the same for one half of the layers: part = TEF_NET_ENCODERS (heads + ConvGRU gates), TEF_NET_DECODERS (residual blocks,
 * decoders, prediction heads) or both.  A data-parallel caller reduces the decoder half first and all-reduces its slice
 * of the gradient bucket while the encoder half is still being reduced (train.Trainer). */
/* workspace bytes tef_net_window_wgrads needs in plan->wgrad_ws for `npass` passes when plan->copy_batch is set (0: none) */
size_t tef_net_window_wgrads_workspace(const tef_net_plan *p, int npass);
int tef_net_window_wgrads_part(const tef_net_plan *p, int part, int npass, const float *const *x,
                               const float *const *const *states_in, const float *const *tape, const float *const *gtape,
                               const unsigned long long *ran, void *stream);

/* ---- optimiser step of the training window on flat buffers (train_flow.py:127-131) ----------------------------------
 * tef_l2_norm: out[0] = ||x||_2 (double partial sums in `scratch`, tef_l2_norm_scratch_bytes, fixed summation order);
 * step (optional) is incremented by one: the counter of the Adam update that follows.
 * tef_adam_clip_step: clip_grad_norm_ (g *= min(1, max_norm / (norm + 1e-6)); max_norm <= 0: no clipping), torch.optim.Adam
 * with amsgrad=False / weight_decay=0 (exp_avg m, exp_avg_sq v, bias corrections from the device counter `step`, which
 * makes the call graph-capturable) and zero_grad (g <- 0), one launch over n elements.  The hyper-parameters are doubles —
 * Python floats in the reference — rounded to fp32 where torch rounds them (1 - beta, lr / bc1, sqrt(bc2)); a NaN norm
 * propagates into every parameter, like torch's clamp. */
size_t tef_l2_norm_scratch_bytes(void);
int tef_l2_norm(const float *x, size_t n, void *scratch, float *out, float *step, void *stream);
int tef_adam_clip_step(float *p, float *g, float *m, float *v, size_t n, const float *norm, float max_norm, double lr,
                       double beta1, double beta2, double eps, const float *step, void *stream);
/* The same with the hyper-parameters read from DEVICE memory at run time, hp = {lr, beta1, beta2, eps, max_norm} (doubles;
 * max_norm <= 0: no clipping): a training window captured in a hipGraph follows a learning-rate schedule — the host
 * refreshes the five numbers before a replay (an ordinary copy, outside the graph) — instead of being captured again. */
int tef_adam_clip_step_hp(float *p, float *g, float *m, float *v, size_t n, const float *norm, const double *hp,
                          const float *step, void *stream);

/* ---- validation metrics (loss/flow_val.py; evaluation only, batch 1, no gradients) ---------------------------------
 * Flow maps are planar [H][W] (fx, fy separately); event lists are loc [N][2] = (y, x), ts [N], mask [N][2]. */
/* one warping step: flow lookup at loc (flow_out [N][2] = (f_y, f_x) if not NULL); if do_warp: loc += (tref - ts) * flow,
 * purge_unfeasible (loc and mask zeroed when out of bounds), ts = tref.  flow_val.py:337-342, :492-517, :528-556 */
int tef_val_event_step(const float *fx, const float *fy, int H, int W, float *loc, float *ts, float *mask, int N,
                       float tref, int do_warp, float *flow_out, void *stream);
/* per-polarity image of an event list, nearest pixel (round_idx, metrics) or bilinear: cnt [2][H][W] (+ tsum [2][H][W]
 * weighted by ts when both are given).  utils/iwe.py:63-136 as used by flow_val.py:129-143, :174-187, :189-274 */
int tef_val_event_image(const float *loc, const float *mask, const float *ts, int N, int H, int W, int round_idx,
                        float *cnt, float *tsum, void *stream);
/* out2 = (FWL, RSAT) from the warped and un-warped count / timestamp images.  flow_val.py:189-274
 * scratch: tef_val_metrics_scratch_bytes(H, W) bytes of device memory (per-workgroup fp64 partial sums). */
size_t tef_val_metrics_scratch_bytes(int H, int W);
int tef_val_metrics(const float *cnt_fw, const float *ts_fw, const float *cnt_zero, const float *ts_zero, int H, int W,
                    float passes, float *out2, void *scratch, size_t scratch_bytes, void *stream);
/* forward propagation of one flow map by dt (scratch3 = 3*H*W floats).  flow_val.py:43-74 */
int tef_val_forward_prop_flow(const float *fx, const float *fy, int H, int W, float dt, float *scratch3, float *out_x,
                              float *out_y, void *stream);
/* accumulated backward flow: indices [2][H][W] (y, x) walk along the newest flow.  flow_val.py:582-604 */
int tef_val_accum_flow(const float *fx, const float *fy, int H, int W, float *indices, float *out_mask, float *acc_x,
                       float *acc_y, void *stream);
/* per-pixel average of P maps over the passes with non-zero flow, optional element-wise divisor [H][W] and event mask
 * [mask_passes][H][W]; out [2][H][W] = (x, y).  flow_val.py:145-172 */
int tef_val_average_flow(const float *maps_x, const float *maps_y, int P, int H, int W, const float *divisor,
                         const float *event_mask, int mask_passes, float *out, void *stream);
/* one-shot per-polarity image of warped events for visualisation: compute_pol_iwe / deblur_events, utils/iwe.py:139-257.
 * flow [B][2][H][W] (ch0 = x), event_list [B][N][4] (ts in [0,1], y, x, p), pol_mask [B][N][2]; out [B][2][H][W]. */
int tef_pol_iwe(const float *flow, const float *event_list, const float *pol_mask, int B, int N, int H, int W,
                int round_idx, int round_flow, float *out, void *stream);
/* Stand-alone forms of the interpolation primitives for callers that use them one by one (utils/iwe.py:63-113
 * get_interpolation, :116-136 interpolate; forward forms — their gradients are the *_backward entry points below).
 * loc [B][n][2] = (y, x).  Bilinear: idx / weights [B][4n] (corner blocks TL, TR, BL, BR along the event axis, linear pixel
 * index as a float like the reference, 0 with weight 0 outside the frame); round_idx: [B][n], nearest pixel, weight 1 / 0.
 * tef_scatter_add: out [B][HW] = 0, then out[b][idx] += weights (* mask, [B][n'] or NULL). */
int tef_interp_corners(const float *loc, int B, int n, int H, int W, int round_idx, float *idx, float *weights, void *stream);
int tef_scatter_add(const float *idx, const float *weights, const float *mask, int B, int n, int HW, float *out, void *stream);
/* The same primitives as DIFFERENTIABLE operators (csrc/tef_prims.hip) for callers that build a loss of their own from
 * utils/iwe.py's functions (the training loss itself runs the fused kernels behind tef_loss_forward / _backward):
 *   tef_event_flow            get_event_flow (utils/iwe.py:17-40) for a whole batch: maps fx, fy [B][H][W], loc [B][N][2] =
 *                             (y, x) -> out [B][N][2] = (f_y, f_x); bilinear, align_corners, taps outside the frame read 0
 *   tef_event_flow_backward   gout [B][N][2] -> dfx, dfy [B][H][W] (ACCUMULATED with float atomics: zero them first; both
 *                             or neither) and dloc [B][N][2] (the lookup's position Jacobian; NULL to skip)
 *   tef_interp_corners_backward  get_interpolation (:63-113), bilinear branch: gweights [B][4n] -> dloc [B][n][2]; autograd
 *                             through max(0, 1 - |d|) as torch runs it (slope halved exactly at |d| = 1, abs'(0) = 0, corners
 *                             outside the frame contribute nothing); indices carry no gradient
 *   tef_scatter_add_backward  interpolate (:116-136): gimg [B][HW] -> dweights = gimg[idx] * mask, dmask = gimg[idx] *
 *                             weights (either NULL to skip; mask NULL = no mask) */
int tef_event_flow(const float *fx, const float *fy, int B, int H, int W, const float *loc, int N, float *out, void *stream);
int tef_event_flow_backward(const float *fx, const float *fy, int B, int H, int W, const float *loc, int N, const float *gout,
                            float *dfx, float *dfy, float *dloc, void *stream);
int tef_interp_corners_backward(const float *loc, int B, int n, int H, int W, const float *gweights, float *dloc, void *stream);
int tef_scatter_add_backward(const float *idx, const float *weights, const float *mask, int B, int n, int HW, const float *gimg,
                             float *dweights, float *dmask, void *stream);
/* average endpoint error over pixels with valid ground truth (and events).  flow_val.py:276-314 */
int tef_val_aee(const float *pred, const float *gt, const float *event_mask, int mask_passes, int H, int W, float *out,
                void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TEF_H */
