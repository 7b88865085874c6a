/*
 * snn_c.h -- C binding of the C++ host mirror (shadernn_amd/host, libsnn_core.so) for tests and bench harnesses.
 *
 * This is NOT the drop-in boundary (that is snnhip.h).  It lets non-C++ callers (pytest via ctypes) drive the same call
 * sequence the reference's own harnesses use:
 *   MixedInferenceCore::create(context, jsonFile, ShaderGenOptions) + run(RunParameters)
 *       -- demo/common/inferenceProcessor.cpp:42-140
 *   ShaderUnitTest::snnConvTestWithLayer: hand-built InputLayer + Conv2D layer, generateInferenceGraph, create, run, read the
 *   "<layer name> pass[0].dump" file            -- demo/common/shaderUnitTest.cpp:174-280
 * All functions return 0 on success; the C++ side aborts (SNN_RIP) on fatal errors exactly like the reference.
 */
#ifndef SNN_C_H
#define SNN_C_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct snn_model snn_model;
struct snnhip_ctx;    /* include/snnhip.h */
struct snnhip_tensor;

/* Loads a JSON model, builds the graph for one W x H x C input image and initialises every stage on `device`.
 * dump_outputs: write "<SNN_OUTPUT_DIR>/<layer name> pass[0].dump" for every layer on each run (disables fusion).
 * fuse_chains : let the HIP backend replace linear runs of layers by fused kernels. */
int snn_model_create(const char* json_path, int device, int in_w, int in_h, int in_c, int dump_outputs, int fuse_chains, int profiling,
                     snn_model** out);
/* as snn_model_create, with the reference's preferHp / ShaderGenOptions::preferrHalfPrecision switch: RGBA16F textures (half tensors in HBM),
 * weights truncated to fp16 at parse time (convertToMediumPrecision), fp16-MFMA convolutions.  Layers without an fp16 kernel (depthwise,
 * dense) abort the run like an unsupported layer does in the reference. */
int snn_model_create2(const char* json_path, int device, int in_w, int in_h, int in_c, int dump_outputs, int fuse_chains, int profiling,
                      int prefer_half, snn_model** out);
/* as snn_model_create2, plus capture_graph: record the first run's launches as a hipGraph and replay it afterwards (ignored with dumps / profiling) */
int snn_model_create3(const char* json_path, int device, int in_w, int in_h, int in_c, int dump_outputs, int fuse_chains, int profiling,
                      int prefer_half, int capture_graph, snn_model** out);
/* as snn_model_create3, for a BATCH of `batch` images per inference: every stage tensor is [batch][H][W][C] (the reference fixes the 4th
 * texture dimension to 1, core/src/ic2/core.cpp:371; here it carries the image count).  Uploads / downloads move batch x H x W x C floats. */
int snn_model_create4(const char* json_path, int device, int in_w, int in_h, int in_c, int dump_outputs, int fuse_chains, int profiling,
                      int prefer_half, int capture_graph, int batch, snn_model** out);
int snn_model_batch(snn_model* m);
/* the C-ABI handles behind a model: its context (device + stream) and the device tensor of its last stage's output (borrowed) */
struct snnhip_ctx* snn_model_hip_ctx(snn_model* m);
struct snnhip_tensor* snn_model_output_tensor(snn_model* m);
int snn_model_destroy(snn_model* m);
int snn_model_upload_input(snn_model* m, const float* nhwc);      /* [batch x] H x W x C floats */
int snn_model_run(snn_model* m);                                   /* MixedInferenceCore::run (enqueue + one sync) */
/* RunParameters::deferSync: enqueue the inference and return; several inferences stay in flight on the model's stream until
 * snn_model_sync (throughput mode; the reference waits once per inference, core.cpp:203) */
int snn_model_run_async(snn_model* m);
int snn_model_sync(snn_model* m);
int snn_model_output_dims(snn_model* m, int hwc[3]);
int snn_model_download_output(snn_model* m, float* nhwc);
/* SNNModelOutput of the next runs (snn.h ModelType: 0 CLASSIFICATION, 1 DETECTION, 2 SEGMENTATION, 3 OTHER; MixedInferenceCore::run, core.cpp:228-237) */
int snn_model_set_type(snn_model* m, int model_type);
int snn_model_classifier_output(snn_model* m);                             /* 1-based arg-max of the last stage, 0 = none */
int snn_model_detections(snn_model* m, float* rows6, int max_rows);         /* YOLO stage output {class, score, x, y, w, h}; returns the row count */
/* Input pre-processing on the device, the demo's call sequence (modelInference.cpp:92-97): decoded 8-bit image (1 / 3 / 4 channels) ->
 * convertToRGBA32FAndNormalize(means, norms) -> resize to the model's input size with (resize_means, resize_norms), bilinear.
 * The model must take a 4-channel input (RGBA texture). */
int snn_model_upload_input_u8(snn_model* m, const unsigned char* pixels, int w, int h, int channels, const float means[4], const float norms[4],
                              const float resize_means[4], const float resize_norms[4]);
int snn_model_num_stages(snn_model* m);
/* *fused_away: bit 0 = the stage's work moved into a later stage's fused plan; bit 1 = the stage is issued on the side stream beside the
 * previous launching stage (SNN_BRANCH_OVERLAP=1) */
int snn_model_stage_info(snn_model* m, int stage, char* name, int name_len, int hwc[3], int* fused_away);
int snn_model_download_stage(snn_model* m, int stage, float* nhwc);
int snn_model_describe(snn_model* m, char* buf, int buflen);
/* The plan a stage launches after fusion (0 steps: input layer, CPU stage, or folded into a later stage's fused plan): kernel description and
 * algorithmic cost of each of its launches, per-launch HIP-event profiling (snnhip_plan_profile_*) and the summed cost of one inference. */
int snn_model_stage_plan_steps(snn_model* m, int stage);
int snn_model_stage_plan_step(snn_model* m, int stage, int step, char* desc, int desc_len, double* flops, double* bytes);
int snn_model_profile_enable(snn_model* m, int enable);
int snn_model_profile_read(snn_model* m, int stage, int step, double* total_ms, int* launches);
/* run launch by launch instead of replaying the recorded hipGraph (1) / replay again (0): what a launch trace (snnhip_trace_begin) needs --
 * a replayed graph never calls the plans */
int snn_model_suspend_replay(snn_model* m, int suspend);
int snn_model_cost(snn_model* m, double* flops, double* bytes);
/* per-stage device timers of the last run, milliseconds (MixedInferenceCore::writeTimeStat); returns count written */
int snn_model_time_stats(snn_model* m, char* names, int names_len, double* ms, int max_entries);

/* ---- multi-device pool (shadernn_amd/host/pool.cpp; SURVEY 8b "Threading" + 8e): one replica = one host thread + HipContext + stream per entry of
 * devices[] (an entry may repeat), each owning one MixedInferenceCore per micro-batch slot of its share [g*B/G, (g+1)*B/G) of the global batch.
 * Weights are loaded per replica, nothing crosses devices on the data path.  micro_batch 0 = a replica's share in one pass.
 * The reference is single-device (vulkanBackend.cpp:30-31); its benchmark loop (inferenceProcessor.cpp:84-86) is what one replica runs. */
typedef struct snn_pool snn_pool;
int snn_pool_create(const char* json_path, const int* devices, int n_devices, int in_w, int in_h, int in_c, int prefer_half, int capture_graph,
                    int global_batch, int micro_batch, snn_pool** out);
int snn_pool_destroy(snn_pool* p);
int snn_pool_replicas(snn_pool* p);
int snn_pool_shard(snn_pool* p, int replica, int* first_image, int* images, int* slots);
int snn_pool_output_dims(snn_pool* p, int hwc[3]);
int snn_pool_upload_input(snn_pool* p, const float* nhwc);              /* global_batch x H x W x C floats; every replica takes its images */
/* all replicas released together; each enqueues `steps` passes over its slots (deferSync) and waits once; returns when the slowest is done
 * (seconds = that wall time: the in-process form of bench.py's barrier + MAX over ranks) */
int snn_pool_run(snn_pool* p, int steps, double* seconds);
int snn_pool_download_output(snn_pool* p, float* nhwc);                 /* gather through host memory: global_batch x outH x outW x outC floats */
/* the edge collective of SURVEY 8e: one ncclAllGather per slot over the replicas' streams (librccl.so loaded at run time), rank 0's copy handed to
 * the host.  -3: the replicas do not have equal shares or two of them share a device (RCCL needs one rank per device); -4: RCCL unavailable / failed */
int snn_pool_allgather_output_rccl(snn_pool* p, float* nhwc_rank0);

/* Restatement of ShaderUnitTest::snnConvTestWithLayer (shaderUnitTest.cpp:174-280): input HWC floats, weights as OC*IC
 * matrices of k x k (flat OIHW), bias[OC], optional BN (4 arrays of OC or NULL), pad: 0 constant 1 replicate 2 reflect.
 * Writes the layer dump and returns its path in dump_path; the dump is the reference's ".dump" format. */
int snn_conv_test_with_layer(int device, const float* input_hwc, const float* weights_oihw, const float* bias, int width, int height, int in_channels,
                             int out_channels, int kernel, int stride, int pad, int use_bn, const float* bn_gamma, const float* bn_mean,
                             const float* bn_var, const float* bn_beta, char* dump_path, int dump_path_len);

/* Host-only (no GPU): parse the JSON model, build the layer DAG and the inference graph for a W x H x C input and print one
 * line per layer: "<index>|<name>|<exec type>|<out W>x<out H>x<out C>|<inputs>".  Exercises ModelParser, layerFactory,
 * topological sort and the shape rules exactly as MixedInferenceCore::create would. */
int snn_graph_summary(const char* json_path, int in_w, int in_h, int in_c, char* buf, int buflen);

/* Host-only: YOLOLayer's decode + NMS (yololayer.cpp:114-226) on two NHWC heads of net/32 and net/16 cells x 18 channels */
int snn_yolo_decode(const float* head_coarse, const float* head_fine, int net_size, float* rows6, int max_rows);

/* Host-only: parse `text` as ONE JSON number with the model loader's own parser (ic2/json.h); returns 0 and the value, -1 if it is not a number.
 * Test hook for the loader's number grammar (overflow saturates to +-HUGE_VAL, underflow gives +-0 / a denormal, like the reference's strtod). */
int snn_json_number(const char* text, double* out);

/* ".dump" reader (image.cpp:300-311): returns W,H,D,C and, if out != NULL, the RGBA32F pixels ([D][H][W][4] floats). */
int snn_dump_read(const char* path, int whdc[4], float* out, long out_floats);

#ifdef __cplusplus
}
#endif
#endif
