/*
 * roomnet_hip.h -- C ABI of libroomnet_hip.so, the MI355X (gfx950) implementation
 * of RoomNet's forward-pass inference path.
 *
 * The reference (ironhide23586/RoomNet) has no native / FFI interface: its
 * boundary is the Python API of network.RoomNet, which hands the whole forward
 * pass to TensorFlow through tf.Session.run.  This library replaces exactly
 * that hand-off.  Each entry point names the reference interface it replaces
 * (file:line into the reference tree); INTEGRATION.md shows the ctypes binding
 * a maintainer of the reference would add to network.py.
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer it passes in; the
 *     library owns its device memory, stream(s), events and packed weights.
 *   - every function returning int returns RN_OK (0) or a negative RN_E_* code;
 *     rn_last_error() returns a thread-local message for the last failure.
 *   - a handle is bound to one device and one stream; calls on one handle must
 *     be serialised by the caller (the reference is single-threaded,
 *     infer.py:79-82); distinct handles are independent.
 *   - tensors are NHWC, C-contiguous.  Images are [n, S, S, 3].
 */
#ifndef ROOMNET_HIP_H
#define ROOMNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RN_API __attribute__((visibility("default")))
#else
#define RN_API
#endif

#define RN_OK 0
#define RN_E_INVALID (-1)   /* bad argument / unsupported graph                     */
#define RN_E_HIP (-2)       /* a HIP runtime call failed (message has the details) */
#define RN_E_NOMEM (-3)     /* host or device allocation failed                    */
#define RN_E_STATE (-4)     /* call not valid in the handle's current state        */
#define RN_E_RANGE (-5)     /* n exceeds max_batch, bad node id, buffer too small  */

/* storage / MFMA input type of activations and conv weights.  float32 everywhere on
 * RN_DTYPE_F32 handles: without RN_FLAG_TAPS their conv stages 1-6 run as one launch each on the
 * matrix cores (v_mfma_f32_32x32x2_f32: float32 in, float32 accumulate; rn_stage_f32m.hip), with
 * RN_FLAG_TAPS as one launch per graph node (every node readable); the two differ only in the order
 * of the convolution's K sum and of the pooling window sum (<= 2e-5 of a tensor's abs-max).
 * On the 16-bit handles: conv accumulation, ReLU6, BN, the vertical
 * part of the residual interpolation, the dense head and the softmax are float32; what is
 * NOT float32 (all inside the tolerances of tests/test_hip_fused.py):
 *   - stage 0 runs on fp16 MFMAs in BOTH 16-bit modes and is exact up to the float32 summation
 *     order: the operand is the uint8 pixel value itself, the pre-processing of network.py:129
 *     is folded into the weights, and every folded weight is an fp16 hi + lo pair in two idle
 *     rows of the 32 x 32 tile (rn_stage.h, s0_pixel_halves); only its OUTPUT is rounded to
 *     the storage type;
 *   - avg-pool 4x4 stride 1 (stages 1-3): ReLU6 outputs are rounded to fp16, vertical pair
 *     sums are fp16 adds, and the window sums run on the matrix cores (fp16 x 0/1 band
 *     matrix, exact float32 accumulation); stride-2 pools are float32 VALU sums, except
 *     stages 4 and 5 wherever the row-blocked kernels run (rn_stage4x.hip / rn_stage5x.hip: rows cut
 *     into column blocks of 194-206 / 66-110 input columns -- 224, 420 and 600 inputs all qualify),
 *     which pool like the stride-1 stages (fp16 ReLU6 outputs and pair sums, band-matrix MFMA);
 *   - every stage that pools fp16 values on the matrix cores (stages 1-3 always, 4 and 5 where
 *     the row-blocked kernels run) keeps its conv weights DIVIDED BY 6 (rounded to the 16-bit
 *     type after the division): relu6(6 x) / 6 = clamp(x, 0, 1) is then the free clamp of the
 *     fp16 conversion, and the folded BN scale carries the 6;
 *   - residual resize: the horizontal interpolation is an MFMA against the interpolation
 *     matrix in the storage type -- stage 3, and stage 5 on the row-blocked kernel: one operand, lerp
 *     fraction rounded to 2^-8 (bf16) / 2^-11 (fp16) so that both weights are exact;
 *     stage 9, and stage 5 on the round-2 kernel (RN_FLAG_PAIR_32X32): hi + lo split (~16-bit weights);
 *   - rounding (round 6, default; RN_FLAG_NO_DITHER restores plain rounding): the conv weights of stages 1-9 are converted to
 *     the storage type with the rounding residual CARRIED from tap to tap of every (cin, cout) pair -- a weight may be one ulp
 *     from its nearest value, the nine taps sum to the exact sum within half an ulp -- and bf16 handles store the outputs of
 *     stages 3, 4, 5 through v_cvt_sr_bf16_f32 with a seed that depends on the output row (1/6, 3/6, 5/6 of an ulp for rows
 *     0, 1, 2 mod 3): deterministic, each stored value within one ulp of the exact one;
 *   - channels rn_create proves constant are not computed at all (rn_frozen_info, rn_const_info): same bits as computing them. */
#define RN_DTYPE_F32 0      /* reference arithmetic type (TensorFlow float32)      */
#define RN_DTYPE_BF16 1
#define RN_DTYPE_F16 2

/* rn_create flags */
#define RN_FLAG_TAPS 1u     /* unfused per-node path; every graph node can be read
                               back with rn_tap (float32 only)                     */
#define RN_FLAG_STAGE_LAUNCHES 2u /* 16-bit handles: one launch per conv stage, every
                               stage output "sK.bn"/"sK.bn2" materialised in HBM (per-stage
                               parity taps).  Default: stages are fused across their
                               boundaries where a kernel exists (the output of a stage
                               that only feeds its fused successor is then never written:
                               s0.bn, s2.bn; and, when a call carries at least half as many images
                               as the device has CUs, s6.bn and s7.bn -- the back end then runs
                               as one launch per image, rn_backend.hip; rn_tap refuses them) */
#define RN_FLAG_GENERIC_KERNELS 4u /* 16-bit handles: every stage on the generic
                               stage_mfma_kernel (diagnostic cross-check of the tuned kernels) */
#define RN_FLAG_COMPUTE_FROZEN 16u /* convolve every channel.  By default channels that rn_create PROVES constant (the stored
                                    * value is the same number for every possible input: BN gammas the reference's L2
                                    * regulariser drove to ~1e-20) are written from a table instead of being convolved and
                                    * enter the next convolution as one constant per cout -- 16-bit handles (the fused
                                    * pair's on-chip tensor, the 64 -> 64 residual stage) and float32 matrix-core handles
                                    * (the same stages) alike; this flag is the comparison arm that shows it.  See
                                    * rn_frozen_info, rn_const_info (the 16 constant output channels of the first step of
                                    * the 64-channel block, round 6, are computed under this flag too). */
#define RN_FLAG_NO_DITHER 32u /* 16-bit handles: plain rounding everywhere -- conv weights rounded to nearest one by one and every
                                store round-to-nearest-even, as in rounds 1-5.  Default (round 6): the weights' rounding residual
                                is carried from tap to tap (the nine taps of a (cin, cout) pair sum to the exact sum within half an
                                ulp), and bf16 handles store the outputs of stages 3, 4, 5 through v_cvt_sr_bf16_f32 with a
                                seed that depends on the output row -- a deterministic dither, each value within one ulp of the
                                exact one -- so that the rounding errors of a 3 x 3 window cancel instead of adding on smooth
                                image content: max |dlogit| on the parity set 0.080 -> 0.030 (600 x 600: 0.126 -> 0.026; profiles/r6_c_parity.json).
                                Comparison arm. */
#define RN_FLAG_PAIR_32X32 8u /* 16-bit handles: the round-2 kernels instead of the round-3 ones (comparison
                               arm, bench.py --pair32): the cross-stage fused pair of the 32-channel block
                               (network.py:183-203: rn_stage23.hip instead of rn_stage23x.hip; equal up to the fp32 order of
                               the pooling sums: ~1 value in 4e7 differs, by one 16-bit ulp);
                               the first step of the 64-channel block, its residual step and the 128-channel
                               step (network.py:216-235: rn_stage_rw.hip / rn_conv16.hip instead of
                               rn_stage4x/5x/6x.hip; these differ in the last 16-bit place: other
                               accumulation order, pooling of stride-2 steps as fp16 band-matrix MFMAs); the
                               stage-0 fusion with wave-private rings instead of the shared ring (bit-identical).
                               TEST / A-B LIBRARY ONLY (round 6): the round-2 pair kernel is linked into
                               roomnet_amd/lib/libroomnet_hip_ab.so (same exports, csrc/build.sh builds both); the
                               product library answers this flag with RN_E_INVALID */

#define RN_MAX_STAGES 16
#define RN_MAX_DENSE 8
#define RN_NAME_LEN 32

typedef struct rn_handle rn_handle;

/* One conv stage = conv3x3 VALID stride-1 no-bias -> ReLU6 -> [avg-pool k,s VALID]
 * -> BN(inference) -> [ + legacy bilinear resize(output of skip_stage) -> BN ].
 * Mirrors one depth step of conv_block (reference network.py:172-208). */
typedef struct rn_conv_stage {
    int32_t cin, cout;
    int32_t pool_k, pool_s;   /* pool_k == 0: no pooling                            */
    int32_t skip_stage;       /* index of the stage whose output is added, or -1    */
    const float* kernel;      /* HWIO [3,3,cin,cout]         (convN/kernel)         */
    const float* gamma;       /* BN after the pool, [cout]   (batch_normalization_N)*/
    const float* beta;
    const float* mean;
    const float* variance;
    const float* gamma2;      /* BN after the residual add, or NULL                 */
    const float* beta2;
    const float* mean2;
    const float* variance2;
} rn_conv_stage;

/* One dense_block (reference network.py:210-223): x@W [+ bias] -> ReLU6 -> [BN] */
typedef struct rn_dense_layer {
    int32_t nin, nout;
    const float* kernel;      /* [nin, nout]                                        */
    const float* bias;        /* [nout] or NULL                                     */
    const float* gamma;       /* BN params or NULL (all four NULL together)         */
    const float* beta;
    const float* mean;
    const float* variance;
} rn_dense_layer;

/* The restored model: what RoomNet.__init__ + RoomNet.load build and restore
 * (reference network.py:21-48, :105-126).  Pointers are host memory and are
 * only read during rn_create. */
typedef struct rn_weights {
    int32_t im_side;          /* RoomNet(im_side=...)                               */
    int32_t num_classes;      /* RoomNet(num_classes=...)                           */
    int32_t n_stages;
    int32_t n_dense;
    float bn_epsilon;         /* tf.layers.batch_normalization default 1e-3         */
    const rn_conv_stage* stages;
    const rn_dense_layer* dense;
} rn_weights;

/* per-call device timing of the last rn_forward_* (HIP events on the handle's stream) */
typedef struct rn_stage_ms {
    int32_t n_stages;                 /* conv stages timed                          */
    float preprocess_ms;              /* uint8 -> float (only when it is a separate launch) */
    float stage_ms[RN_MAX_STAGES];    /* one fused launch (or launch group) per stage */
    float head_ms;                    /* flatten + dense blocks + softmax + argmax  */
    float total_ms;                   /* first launch -> last launch                */
} rn_stage_ms;

typedef struct rn_node_info {
    char name[RN_NAME_LEN];           /* "s3.conv", "s3.pool", "s3.bn", "s3.add", "s3.bn2", "d0.mm", ... */
    int32_t h, w, c;                  /* per-image shape (h = w = 1 for head nodes) */
} rn_node_info;

/* ---- lifetime ------------------------------------------------------------
 * rn_create replaces RoomNet.__init__(optimized_inference=True) + RoomNet.init()
 * + RoomNet.load(path) (reference network.py:21-48, :87-91, :105-126): it builds
 * the execution plan for `w`, packs/uploads the weights and sizes the workspace
 * for up to max_batch images per call. */
RN_API int rn_create(const rn_weights* w, int device, int dtype, int max_batch, unsigned flags, rn_handle** out);
RN_API void rn_destroy(rn_handle* h);
RN_API const char* rn_last_error(void);
RN_API int rn_device_count(void);
RN_API const char* rn_version(void);

/* ---- forward pass ----------------------------------------------------------
 * rn_forward_u8 replaces RoomNet.infer(im_batch) (reference network.py:128-135):
 * BGR uint8 [n,S,S,3] -> channel flip + ((x/255.)*2)-1 (float64 semantics, then
 * float32) -> graph -> (argmax int64 [n], softmax float32 [n,num_classes]).
 * Host buffers; blocks until the results are in probs/ids. */
RN_API int rn_forward_u8(rn_handle* h, const uint8_t* bgr_nhwc, int n, float* probs, int64_t* ids);

/* Two-slot pipelined form of rn_forward_u8 for a caller that classifies batch after batch from host memory (the
 * directory driver, infer.py:79-82, at batch size): rn_submit_u8 uploads the batch into the slot's device buffer on a
 * copy stream and enqueues its forward pass + result download behind it; rn_collect waits for that slot and copies the
 * results out.  With  submit(0) submit(1) collect(0) submit(0) collect(1) ...  the upload of one batch overlaps the kernels
 * of the previous one (from pageable memory the upload call itself blocks the calling thread, the GPU does not idle;
 * from pinned memory -- rn_host_alloc -- it is asynchronous: the caller then keeps the slot's source buffer unchanged
 * until rn_collect(slot) has returned, i.e. one pinned buffer per slot).  slot = 0 or 1; a slot must be collected before
 * it is submitted again. */
RN_API int rn_submit_u8(rn_handle* h, const uint8_t* bgr_nhwc, int n, int slot);
RN_API int rn_collect(rn_handle* h, int slot, float* probs, int64_t* ids);

/* rn_forward_f32 replaces sess.run(outs_final, {x_tensor: im}) (reference
 * network.py:133/:155) for an already pre-processed RGB float32 batch in [-1,1]. */
RN_API int rn_forward_f32(rn_handle* h, const float* rgb_nhwc, int n, float* probs, int64_t* ids);

/* Device-resident variants: all pointers are device memory on the handle's
 * device; the call only enqueues work on the handle's stream (asynchronous).
 * Use rn_sync (or your own event on the stream) before reading the outputs. */
RN_API int rn_forward_u8_device(rn_handle* h, const uint8_t* d_bgr_nhwc, int n, float* d_probs, int64_t* d_ids);
RN_API int rn_forward_f32_device(rn_handle* h, const float* d_rgb_nhwc, int n, float* d_probs, int64_t* d_ids);
/* The caller's image pipeline on the device -- RoomNet.center_crop + cv2.resize(INTER_LINEAR) of
 * RoomNet.infer_optimized (reference network.py:137-146, :152).  `d_src` is one BGR uint8 HWC image of any size
 * in device memory; its centred square window is resized to im_side x im_side into slot `index` of a device batch
 * buffer [max_batch, im_side, im_side, 3] that rn_forward_u8_device then takes.  Integer-exact restatement of
 * OpenCV's fixed-point algorithm (bit-identical to roomnet_amd/imageops.py).  Asynchronous on the handle's stream. */
RN_API int rn_crop_resize_u8_device(rn_handle* h, const uint8_t* d_src, int src_h, int src_w, uint8_t* d_dst_batch, int index);
/* The same for a BATCH of device-resident images in ONE launch: image i (d_srcs[i], heights[i] x widths[i] BGR uint8, HWC, any
 * sizes) is centre-cropped and resized into slot i of d_dst_batch ([n, S, S, 3]).  Asynchronous on the handle's stream;
 * n <= max_batch.  network.py:149-152 for a whole directory's worth of decoded images at once. */
RN_API int rn_crop_resize_batch_u8_device(rn_handle* h, const uint8_t* const* d_srcs, const int* heights, const int* widths, int n,
                                          uint8_t* d_dst_batch);
/* infer.py:79-82 for a whole batch: n host images of individual sizes heights[i] x widths[i] (BGR uint8 HWC) are
 * uploaded, centre-cropped, resized and classified; probs [n, num_classes], ids [n] on the host.  Synchronous. */
RN_API int rn_classify_images_u8(rn_handle* h, const uint8_t* const* images, const int* heights, const int* widths, int n,
                                 float* probs, int64_t* ids);
RN_API int rn_sync(rn_handle* h);

/* Run on a caller-provided hipStream_t (e.g. the framework's current stream)
 * instead of the handle's own; NULL restores the handle's stream (a non-blocking stream,
 * NOT ordered against the HIP null stream).  rn_set_stream_null selects the HIP null
 * (legacy default) stream itself, which a NULL argument cannot express: use it when the
 * caller's other work is on the default stream and must be ordered with the library's. */
RN_API int rn_set_stream(rn_handle* h, void* hip_stream);
RN_API int rn_set_stream_null(rn_handle* h);

/* ---- introspection -----------------------------------------------------------
 * rn_tap copies graph node `node_id` of the last forward call to host float32
 * (layout [n, h, w, c]); needs RN_FLAG_TAPS for conv/pool/add nodes; the
 * tensors a launch WRITES are always available: every stage's output node
 * ("sK.bn", for the residual stages "sK.bn2") unless the stage is fused into
 * its successor's launch.  The first BN output "sK.bn" of a residual stage
 * exists only where that stage runs one launch per graph node (float32 handles
 * with RN_FLAG_TAPS, and stages the matrix-core float32 kernels do not cover);
 * 16-bit handles never materialise it.  Asking for a node that was not written
 * returns RN_E_STATE.  This is the per-layer
 * debug read-out the reference gets from self.layers (network.py:30, :207). */
/* What rn_create folded on this handle (zero / -1 where nothing is): info[0] = channels of the first 32 -> 32 stage's output
 * (16-bit handles: the fused pair's on-chip tensor) that are provably constant and therefore not contracted by the next stage
 * (24, 16 or 0; 16-bit handles since round 6 prove it on the tensor's 16-BIT STORE -- the two ends of the pooled sum's range store
 * the same number -- which includes every channel whose fma returns its addend in float32, the criterion of the float32
 * matrix-core handles: 16 or 0 there), info[1] = how many of its 32 channels were proven so,
 * info[2] = index of the 64 -> 64 residual stage whose frozen first-BN channels are folded (or -1), info[3] = 16-cout
 * quarters of that stage whose convolution still runs (4 = all).  RN_FLAG_COMPUTE_FROZEN and RN_FLAG_TAPS handles report
 * nothing folded. */
RN_API int rn_frozen_info(const rn_handle* h, int info[4]);
/* Constant channels nobody computes (16-bit handles; zero / -1 where nothing is): info[0] = index of the conv stage whose last 16
 * output channels are CONSTANTS in this handle's 16-bit store -- rn_create proves per channel that the store of fma(H, sc, sh) is
 * one 16-bit number for every pooled sum H the stage can produce -- and are therefore not convolved but written once, at
 * rn_create (or -1), info[1] = how many channels of that stage were proven so, info[2] = channels folded (16 or 0), info[3] =
 * input channels the stage behind it still contracts (48; its own last 16 output channels are constants as well).  The shipped
 * checkpoint: stage 4 (network.py:228, first step), 26 channels in bf16, 23 in fp16.  RN_FLAG_COMPUTE_FROZEN handles and
 * float32 handles report nothing. */
RN_API int rn_const_info(const rn_handle* h, int info[4]);
RN_API int rn_node_count(const rn_handle* h);
RN_API int rn_node_info_get(const rn_handle* h, int node_id, rn_node_info* out);
RN_API int rn_tap(rn_handle* h, int node_id, float* out, size_t cap_elems, size_t* n_elems);

/* Enable/disable per-stage event timing (adds event records to every forward). */
RN_API int rn_set_profiling(rn_handle* h, int enable);
RN_API int rn_timing(rn_handle* h, rn_stage_ms* out);

/* Name of the kernel that dominates the forward pass of this handle (for
 * matching rocprofv3 rows) and the stage index it belongs to. */
RN_API int rn_dominant_stage(const rn_handle* h);
/* Launch grouping of the conv stages: the index of the stage under which the launch that
 * computes `stage` reports its time in rn_stage_ms (== stage when the stage has a launch of
 * its own; the last stage of the group when stages are fused across their boundary --
 * the depth loop of conv_block, reference network.py:183-203, is then one kernel).
 * Returns a negative code for a bad argument. */
RN_API int rn_stage_launch(const rn_handle* h, int stage);

/* ---- several GPUs from one process ------------------------------------------------
 * The reference is one tf.Session on one device (network.py:89).  A group is one rn_handle
 * per device; a batch is split contiguously (device d gets images [d*n/N, (d+1)*n/N), the
 * first n % N devices one more) and the ONLY exchange on the data path is one RCCL
 * all-gather of every device's packed results over xGMI: per device `slot_bytes` =
 * max_batch_per_device * (num_classes * 4 + 8) bytes = probs [cap, C] float32 followed by
 * ids [cap] int64 (32 bytes per image).  librccl is loaded on the first rn_group_create
 * (environment ROOMNET_RCCL_LIB, read at that call only: another file to dlopen instead);
 * without it rn_group_create returns RN_E_STATE and says which dlopen failed.
 * One host thread drives all devices; calls on one group must be serialised by the caller;
 * every rn_group_* call leaves the caller's current HIP device as it found it.
 * SCALING: rn_group_forward_u8_device is the entry that scales -- the shards are already in
 * each device's HBM (as in BASELINE's measurement contract) and nothing but 32 B per image
 * crosses a link.  rn_group_forward_u8 takes one host buffer: it uploads the shards through
 * one persistent host thread per device (started by rn_group_create, parked between calls:
 * a copy out of pageable memory blocks the thread that issues it; out of an rn_host_alloc
 * buffer it does not) and is bound by the host's memory and PCIe bandwidth (38.5 MB per
 * device and call at 256 x 224 x 224), not by the GPUs.
 * VALIDATION: groups of more than one device have not run on hardware yet (the
 * development pool has one MI355X per box); the one-device group is tested on the GPU. */
typedef struct rn_group rn_group;
/* devices == NULL: devices 0 .. ndev-1.  Weights are replicated (0.7 MB). */
RN_API int rn_group_create(const rn_weights* w, int ndev, const int* devices, int dtype, int max_batch_per_device,
                           unsigned flags, rn_group** out);
RN_API void rn_group_destroy(rn_group* g);
RN_API int rn_group_size(const rn_group* g);
RN_API rn_handle* rn_group_handle(rn_group* g, int index);     /* the per-device handle (profiling, taps) */
/* RoomNet.infer(im_batch) (reference network.py:128-135) over all devices: host BGR uint8
 * [n,S,S,3] in, (softmax [n,C], argmax [n]) on the host out; n <= ndev * max_batch_per_device.
 * Blocks until the results are in probs/ids. */
RN_API int rn_group_forward_u8(rn_group* g, const uint8_t* bgr_nhwc, int n, float* probs, int64_t* ids);
/* Device-resident form: d_shards[d] = device d's images already in its HBM, counts[d] of them.
 * Asynchronous: enqueues the forward passes and the all-gather on the devices' streams;
 * afterwards rn_group_result_buffer(g, d, ...) on ANY device holds all devices' packed results
 * ([ndev][slot_bytes], device d's slot at d * slot_bytes).  rn_group_sync waits for all of it.
 * A device with counts[d] == 0 runs nothing but still takes part in the all-gather: its slot
 * keeps what its last non-empty call left there (zeros before the first); only the first
 * counts[d] entries of a slot belong to this call. */
RN_API int rn_group_forward_u8_device(rn_group* g, const uint8_t* const* d_shards, const int* counts);
RN_API int rn_group_result_buffer(rn_group* g, int index, void** d_gathered, size_t* slot_bytes);
RN_API int rn_group_sync(rn_group* g);
/* The shard / slot plan rn_group_forward_u8 applies, as a pure function (no group, no device -- testable on any host): device d
 * of ndev takes the images [offsets[d], offsets[d] + counts[d]) of an n-image batch -- contiguous shards, the first n % ndev
 * devices one image more (network.py:128-135 split over devices; the same rule as roomnet_amd/parallel.py: shard_bounds) --
 * and *slot_bytes (may be null) = the size of one device's slot in the gathered buffer: probs [max_batch_per_device, C] float32
 * followed by ids [max_batch_per_device] int64.  n > ndev * max_batch_per_device is RN_E_RANGE. */
RN_API int rn_group_plan(int n, int ndev, int max_batch_per_device, int num_classes, int* counts, int* offsets, size_t* slot_bytes);

/* ---- simple device memory helpers (so a host language without a HIP binding
 * can keep batches resident in HBM) ---------------------------------------- */
RN_API int rn_device_malloc(rn_handle* h, size_t bytes, void** d_ptr);
RN_API int rn_device_free(rn_handle* h, void* d_ptr);
RN_API int rn_memcpy_h2d(rn_handle* h, void* d_dst, const void* src, size_t bytes);
RN_API int rn_memcpy_d2h(rn_handle* h, void* dst, const void* d_src, size_t bytes);

/* ---- pinned host memory ------------------------------------------------------------
 * The reference hands TensorFlow pageable NumPy arrays (network.py:133); a copy out of pageable
 * memory blocks the calling thread while the runtime stages it.  A batch buffer from
 * rn_host_alloc (page-locked, usable with every device of the process) makes the upload of
 * rn_submit_u8 / rn_forward_u8 / rn_group_forward_u8 a true asynchronous DMA: the two-slot
 * pipeline then hides it completely behind the previous batch's kernels.  Free with
 * rn_host_free (NULL is a no-op).  No handle is needed: call before or after rn_create. */
RN_API int rn_host_alloc(size_t bytes, void** ptr);
RN_API int rn_host_free(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* ROOMNET_HIP_H */
