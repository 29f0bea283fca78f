/*
 * rpw.h -- C-ABI of the MI355X-native Recursive Patchwork ground-segmentation engine.
 *
 * This is the drop-in boundary for ONE hot path of the reference package
 * (PhilipNaveen/ROS2-Recursive-Patchwork-Implementation):
 *
 *   recursive_patchwork::RecursivePatchwork::filterGroundPoints
 *       reference src/recursive_patchwork/include/recursive_patchwork.hpp:53-54  (declaration)
 *       reference src/recursive_patchwork/src/recursive_patchwork.cpp:310-426     (definition)
 *
 * and it supersedes the reference's per-op accelerator seam, which crosses the
 * host/device boundary once per op and per fit iteration:
 *
 *   cuda::ops::computeDistances2D / filterPointsByRadius / computeAngles / computePlaneDistances
 *       reference src/recursive_patchwork/include/cuda_interface.hpp:59-88
 *       reference src/recursive_patchwork/cuda/cuda_interface.cu:582-654
 *   cuda_compute_distances_2d / cuda_filter_points_by_radius / cuda_compute_angles /
 *   cuda_classify_points_in_patch / cuda_compute_plane_distances / cuda_classify_ground_points
 *       reference src/recursive_patchwork/cuda/cuda_interface.cu:18-36 (forward declarations)
 *       reference src/recursive_patchwork/cuda/cuda_wrapper.cu:139-212  (launchers)
 *
 * Here the seam is one call per frame (or per batch of frames): plain pointers and
 * sizes in, labels out; the library owns device memory, streams and kernels.
 * Nothing in this header names a torch, Eigen or STL type. No function throws;
 * every function that can fail returns an rpw_status and leaves a message
 * retrievable with rpw_last_error().
 *
 * There is NO CPU fallback behind this ABI: if no HIP device is usable,
 * rpw_create() fails with RPW_ERR_NO_DEVICE.
 */
#ifndef RPW_H
#define RPW_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPW_ABI_VERSION 3

typedef enum rpw_status {
    RPW_OK              = 0,
    RPW_ERR_INVALID     = 1,  /* null pointer, bad size, bad config                      */
    RPW_ERR_NO_DEVICE   = 2,  /* no usable HIP device / kernels not loadable             */
    RPW_ERR_HIP         = 3,  /* a HIP runtime call failed (message has the HIP string)  */
    RPW_ERR_CAPACITY    = 4,  /* batch exceeds the capacity given to rpw_create          */
    RPW_ERR_UNSUPPORTED = 5,  /* config outside the supported range (num_sectors > 128)  */
    RPW_ERR_BUSY        = 6   /* rpw_submit on a pipeline slot whose frame has not been rpw_wait-ed */
} rpw_status;

/* Mirrors recursive_patchwork::PatchworkConfig field for field
 * (reference include/recursive_patchwork.hpp:25-36); same defaults via rpw_default_config().
 * The ring count (8) and r_min (1.0) are hard-coded in the reference
 * (recursive_patchwork.cpp:344-345) and are hard-coded here too. */
typedef struct rpw_config {
    float   sensor_height;        /* 1.2  */
    float   max_range;            /* 150  (never read by the path, kept for layout parity) */
    int32_t num_sectors;          /* 10   */
    int32_t max_iter;             /* 100  */
    int32_t adaptive_seed_height; /* 1    */
    float   th_seeds;             /* 0.15 */
    float   th_dist;              /* 0.2  */
    float   th_outlier;           /* 0.08 (never read by the path) */
    float   filtering_radius;     /* 150  */
    int32_t max_split_depth;      /* 1000 */
} rpw_config;

/* One label per INPUT point. The reference returns two point vectors
 * (recursive_patchwork.cpp:402-425); they are recoverable from the labels:
 *   ground     = points labelled GROUND, input order
 *   non-ground = points labelled NONGROUND (input order), then points labelled
 *                OUTSIDE (input order); DROPPED points appear in neither.
 * If stats.n_zone < 3 the reference returns {{}, cleaned} (recursive_patchwork.cpp:339-341):
 * then non-ground = every point not DROPPED, input order. */
enum {
    RPW_LABEL_NONGROUND = 0,
    RPW_LABEL_GROUND    = 1,
    RPW_LABEL_DROPPED   = 2,  /* non-finite coordinate (cleanPoints, recursive_patchwork.cpp:19-35) */
    RPW_LABEL_OUTSIDE   = 3,  /* finite, sqrt(x^2+y^2) > filtering_radius                           */
    RPW_LABEL_EGO       = 4   /* only with the fusion pre-step: removed by the ego-radius filter, in no cloud */
};

typedef struct rpw_frame_stats {
    uint32_t n_in, n_clean, n_zone, n_ground;
    uint32_t n_patches;          /* non-empty ring x sector patches                   */
    uint32_t n_nodes;            /* fitPlaneAndSplit calls (recursion tree nodes)     */
    uint32_t n_splits;
    uint32_t max_depth;
    uint64_t total_iterations;   /* plane refits inside the loops, summed over nodes  */
    uint64_t chain_addends;      /* elements summed sequentially per accumulator row, over all chained sweeps
                                    (the work the add-dependency chains had to do; not defined by the reference) */
} rpw_frame_stats;

/* Recursion-tree trace record (optional output, see rpw_set_trace). */
enum {
    RPW_NODE_FIT = 0, RPW_NODE_REJECT = 1, RPW_NODE_SMALL = 2, RPW_NODE_FLAT = 3, RPW_NODE_SPLIT = 4
};
typedef struct rpw_node {
    int32_t  frame;
    int32_t  patch;        /* ring * num_sectors + sector                        */
    int32_t  depth;
    uint32_t start, len;   /* slots, relative to the patch's first slot          */
    int32_t  kind;
    int32_t  iterations;
    uint32_t n_inliers;
    int32_t  split_axis;   /* SPLIT: 0 = x, 1 = y; else -1                       */
    float    split_value;  /* SPLIT: the median                                  */
    float    centroid[3];  /* final plane of the node (recursive_patchwork.cpp:228) */
    float    normal[3];
    float    residual;
} rpw_node;

typedef struct rpw_handle rpw_handle;

/* kernels timed by rpw_set_profiling / rpw_get_kernel_ms */
enum {
    RPW_K_BIN_COUNT   = 0,   /* binning pass 1/3: classify, per-tile class counts, frame class totals          */
    RPW_K_SCAN        = 1,   /* binning pass 2/3: per-frame scan -> patch begins, destination of every tile run */
    RPW_K_BIN_SCATTER = 2,   /* binning pass 3/3: classify, stable rank, write the patch-sorted arrays          */
    RPW_K_FIT         = 3,   /* recursive plane fit of every patch                                              */
    RPW_K_COUNT       = 4
};

void        rpw_default_config(rpw_config* cfg);
uint32_t    rpw_abi_version(void);

/* device: HIP device ordinal. max_points: capacity in points of one batch (sum over its
 * frames), at most 2^30 (RPW_ERR_CAPACITY beyond: the kernels address a batch with 32-bit byte
 * offsets); a single frame may hold at most 2^28 - 1 points. max_frames: capacity in frames of
 * one batch. Device memory taken: about 70 bytes per point of capacity. */
rpw_status  rpw_create(const rpw_config* cfg, int device, uint32_t max_points, uint32_t max_frames,
                       rpw_handle** out);
void        rpw_destroy(rpw_handle* h);
rpw_status  rpw_set_config(rpw_handle* h, const rpw_config* cfg);
rpw_status  rpw_get_config(const rpw_handle* h, rpw_config* cfg);
const char* rpw_last_error(const rpw_handle* h);   /* h may be NULL: message of the last failed rpw_create */

/* Host-pointer, one frame, synchronous: the replacement for one filterGroundPoints call.
 *   xyz      n x 3 floats, AoS (Point3D layout), host memory
 *   labels   n bytes out, host memory
 *   residual n floats out or NULL: |(p_i - c).n| against the final plane of the fitted leaf
 *            that labelled point i; -1 for points labelled by an early-out / no patch
 *   stats    out or NULL */
rpw_status  rpw_segment(rpw_handle* h, const float* xyz, uint32_t n, uint8_t* labels,
                        float* residual, rpw_frame_stats* stats);

/* Same call on the bytes of a sensor_msgs/PointCloud2 message (what the reference node receives,
 * reference src/recursive_patchwork_node.cpp:61-88): n = width * height records of point_step bytes with
 * little-endian FLOAT32 x / y / z fields at byte offsets off_x / off_y / off_z (4-byte aligned). The records
 * are copied to the device as they are and read in place by the binning kernel: no host-side repacking. */
rpw_status  rpw_segment_pointcloud2(rpw_handle* h, const uint8_t* data, uint32_t n, uint32_t point_step,
                                    uint32_t off_x, uint32_t off_y, uint32_t off_z,
                                    uint8_t* labels, float* residual, rpw_frame_stats* stats);

/* ---- Host-pointer PIPELINE: the drop-in path with K frames in flight -------------------------------------------
 * Every caller of the reference holds one frame in host memory and calls filterGroundPoints once per frame
 * (reference src/recursive_patchwork_node.cpp:61-108 per PointCloud2 message, src/main.cpp:268,286,
 * test/test_recursive_patchwork.cpp:68,155). rpw_segment is that call, synchronous: one frame's latency (1 - 5 ms on
 * HDL-64 frames, set by the longest sequential add chain of its largest patch) bounds it to a few hundred frames/s. The
 * pipeline keeps K frames in flight instead: every slot owns a HIP stream, pinned staging for the frame and its results,
 * and its own device workspace, so the host copies and the kernels of different slots overlap.
 *
 *   rpw_set_pipeline(h, K)            K slots (0 frees them); each takes the handle's max_points of capacity
 *   rpw_submit(h, slot, xyz, n, f)    copies the frame into the slot's pinned buffer and enqueues the slot's kernels on
 *                                     its stream (kernels only: the binning kernel reads the pinned buffer in place, a
 *                                     copy kernel writes labels / residuals / stats back to pinned memory -- copy-engine
 *                                     transfers made the slots wait for each other); returns at once. RPW_ERR_BUSY if
 *                                     the slot's previous frame has not been waited for. flags: RPW_SUBMIT_RESIDUAL.
 *   rpw_slot_input(h, slot)           the slot's pinned input buffer (max_points x 3 floats): a caller that fills it
 *                                     in place and passes it to rpw_submit saves the host copy. The pointer stays valid
 *                                     until rpw_set_pipeline / rpw_set_config / rpw_destroy, or until an
 *                                     rpw_submit_pointcloud2 on this slot needs more than max_points x 12 bytes and
 *                                     replaces the buffer (ask again after such a call)
 *   rpw_poll(h, slot, &done)          non-blocking: has the slot's frame finished?
 *   rpw_wait(h, slot, labels, residual, stats)   blocks until the slot's frame is done, copies its results out and
 *                                     frees the slot. Slots complete independently and may be waited in any order.
 * One thread drives a handle (as everywhere in this ABI); K is at most RPW_MAX_SLOTS.
 * Streams need hardware queues of their own to overlap: the library sets GPU_MAX_HW_QUEUES=16 when it is loaded
 * unless the variable is already set (ROCm reads it when the HIP runtime starts; its default of 4 makes slots share
 * queues and serialise). */
#define RPW_MAX_SLOTS 32
#define RPW_SUBMIT_RESIDUAL 1u
rpw_status  rpw_set_pipeline(rpw_handle* h, uint32_t n_slots);
uint32_t    rpw_get_pipeline(const rpw_handle* h);
float*      rpw_slot_input(rpw_handle* h, uint32_t slot);
rpw_status  rpw_submit(rpw_handle* h, uint32_t slot, const float* xyz, uint32_t n, uint32_t flags);
/* the same on PointCloud2 message bytes (see rpw_segment_pointcloud2): what the ROS2 node submits */
rpw_status  rpw_submit_pointcloud2(rpw_handle* h, uint32_t slot, const uint8_t* data, uint32_t n, uint32_t point_step,
                                   uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t flags);
rpw_status  rpw_poll(rpw_handle* h, uint32_t slot, int* done);
rpw_status  rpw_wait(rpw_handle* h, uint32_t slot, uint8_t* labels, float* residual, rpw_frame_stats* stats);

/* ---- LidarFusion pre-step, fused into the binning kernel's load stage --------------------------------------
 * reference src/recursive_patchwork/src/lidar_fusion.cpp:42-86 (fuseLidarPointClouds), :88-107 (processSingleLidar),
 * :109-126 (applyRotation2D), :148-159 + :184-187 (removeEgoVehicle). A frame is then the concatenation of the
 * RAW clouds of the configured sensors; every point is yaw-rotated by its sensor's angle (when |angle| > 1e-6
 * degrees) and dropped when sqrtf(x*x + y*y) <= ego_radius (label RPW_LABEL_EGO); the remaining points, in
 * sensor order, are exactly the cloud fuseLidarPointClouds returns, and they are segmented as usual
 * (stats.n_in counts them). Residuals and the recursion trace refer to the rotated points. */
typedef struct rpw_sensor {
    float rotation_deg;   /* LidarConfig::rotation_angle (recursive_patchwork.hpp:41) */
    float ego_radius;     /* LidarConfig::ego_radius     (recursive_patchwork.hpp:42) */
} rpw_sensor;
rpw_status  rpw_set_sensors(rpw_handle* h, uint32_t n_sensors, const rpw_sensor* sensors);   /* n_sensors <= 8; 0 = off */
/* one frame, host pointers: sensor k's raw cloud is xyz[sensor_offsets[k] .. sensor_offsets[k+1]) */
rpw_status  rpw_segment_fused(rpw_handle* h, const float* xyz, const uint32_t* sensor_offsets,
                              uint8_t* labels, float* residual, rpw_frame_stats* stats);
/* device-resident batch: sensor_offsets is a HOST array [n_frames][n_sensors + 1], relative to each frame */
rpw_status  rpw_segment_batch_device_fused(rpw_handle* h, const float* d_xyz, const uint32_t* frame_offsets,
                                           const uint32_t* sensor_offsets, uint32_t n_frames, uint8_t* d_labels,
                                           float* d_residual, void* stream);

/* Device-resident batch, asynchronous on `stream` (a hipStream_t, NULL = default stream).
 *   d_xyz          device pointer, frames concatenated, AoS
 *   frame_offsets  HOST array of n_frames + 1 point offsets (frame f = [off[f], off[f+1]))
 *   d_labels       device pointer, off[n_frames] bytes
 *   d_residual     device pointer or NULL
 * Returns after enqueueing; results are valid once `stream` has completed. The handle's workspace holds ONE batch
 * at a time: a call on a different stream than the previous one first makes that stream wait (on the device) for
 * the previous batch, so two batches never share scratch memory; use one handle per batch in flight to overlap. */
rpw_status  rpw_segment_batch_device(rpw_handle* h, const float* d_xyz, const uint32_t* frame_offsets,
                                     uint32_t n_frames, uint8_t* d_labels, float* d_residual,
                                     void* stream);
/* Waits for the last batch and copies its per-frame stats (n_frames records) to the host. */
rpw_status  rpw_get_batch_stats(rpw_handle* h, rpw_frame_stats* stats, uint32_t n_frames);

/* Per-frame result record of the last batch, written on the DEVICE as four int32 per frame -- ground points, non-ground
 * points (n_clean - n_ground: the zone's non-ground points plus the cleaned points beyond filtering_radius, i.e. the size
 * of the reference's non_ground_points, recursive_patchwork.cpp:415-423), patches, refits -- into d_out
 * (device memory, 4 * n_frames int32). Enqueued on `stream` behind the batch (an event dependency if `stream` is not the
 * batch's own): what the multi-GPU result gather sends, with no host round trip. If the batch ended with patch_fit's
 * error flag set (the condition rpw_get_batch_stats reports as RPW_ERR_HIP), every field of every record is -1. */
rpw_status  rpw_result_record_device(rpw_handle* h, int32_t* d_out, uint32_t n_frames, void* stream);

/* Optional recursion-tree trace of the last call (costs a little time; off by default). */
rpw_status  rpw_set_trace(rpw_handle* h, int enable, uint32_t max_nodes);
rpw_status  rpw_get_nodes(rpw_handle* h, rpw_node* nodes, uint32_t cap, uint32_t* n_nodes);

/* Per-kernel device time, measured with HIP events on the launch stream. */
rpw_status  rpw_set_profiling(rpw_handle* h, int enable);
rpw_status  rpw_get_kernel_ms(rpw_handle* h, double ms[RPW_K_COUNT], uint64_t launches[RPW_K_COUNT], int reset);

/* Test hooks: the device-side restatements of libm atan2f and of the 3x3 eigensolver,
 * evaluated on the DEVICE over arrays held in host memory. */
rpw_status  rpw_debug_angles(rpw_handle* h, const float* x, const float* y, float* angle, uint32_t n);
rpw_status  rpw_debug_eig3(rpw_handle* h, const float* cov9, float* normal3, uint32_t n);
/* Micro-benchmark of the two loops the plane-fit kernel lives in, on a synthetic node of n slots of which inlier_pct %
 * are inliers, one node per workgroup, `blocks` workgroups, `reps` passes each: what = 0..5 chain sweeps (centroid,
 * covariance, residual, range, xy, variance), 16..18 fused / mask passes (seed, plane, older); + 64: the same loop in the
 * chain-server form (seven nodes per server wave), + 96: with a server wave per node. ms = device time of the launch. */
rpw_status  rpw_debug_loop_bench(rpw_handle* h, int what, uint32_t n, uint32_t inlier_pct, int reps, uint32_t blocks, double* ms);
/* The same, and the shader-clock ticks (s_memtime) node 0's owner wave spent in its `reps` passes: the chip's clock follows
 * its recent load, so cycles -- not wall time -- compare the loops of a LONE patch. ticks may be NULL. */
rpw_status  rpw_debug_loop_bench_ticks(rpw_handle* h, int what, uint32_t n, uint32_t inlier_pct, int reps, uint32_t blocks, double* ms, double* ticks);
/* Host evaluation of the same atan2f restatement (no device needed). */
void        rpw_debug_angles_host(const float* x, const float* y, float* angle, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RPW_H */
