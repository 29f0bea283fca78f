// rpw_internal.h -- structures shared between the host orchestration (rpw_capi.hip) and the
// kernels (rpw_kernels.hip). Not part of the public ABI.
#pragma once

#include <stdint.h>
#include "../../include/rpw.h"
#include "rpw_math.h"

namespace rpw {

constexpr int kWave        = 64;
constexpr int kTileThreads = 256;                 // 4 waves
constexpr int kTileItems   = 8;                   // points per thread
constexpr int kTilePoints  = kTileThreads * kTileItems;   // 2048
constexpr int kWavePoints  = kWave * kTileItems;  // 512: one wave owns a contiguous sub-tile
constexpr int kMaxSectors  = 128;
constexpr int kMaxSensors  = 8;                   // LidarFusion pre-step: sensors per frame
constexpr int kBuckets     = 64;                  // patch-size buckets for largest-first launch order
constexpr int kRingFloats  = 2048;                // LDS floats of a patch owner's addend ring (sweeps, fused pass)
constexpr int kStackDepth  = 48;                  // explicit recursion stack (smaller child first => <= log2(n)+2)

struct PatchDesc {
    uint32_t begin;   // absolute slot in the patch-sorted SoA arrays
    uint32_t n;
};

struct P3 { float x, y, z; };

// LidarFusion pre-step folded into the load stage (reference src/lidar_fusion.cpp:88-126,148-159,184-187)
struct SensorTable {
    int   n;                        // 0 = no fusion step
    int   rotate[kMaxSensors];      // |rotation_angle| > 1e-6
    float cosv[kMaxSensors], sinv[kMaxSensors];
    float gt_ego[kMaxSensors];      // min { s : sqrtf(s) > ego_radius }: d <= ego_radius  <=>  x*x + y*y < gt_ego
};

// Everything a kernel needs, passed by value.
struct DevParams {
    rpw_config  cfg;
    PolarTables tab;
    const float* bounds;          // [S+1] sector bounds, device memory
    int S, P, PB;                 // sectors, patches per frame (8*S), histogram row length (P+3, padded)

    uint32_t        n_frames;
    uint32_t        n_tiles;      // over all frames
    uint32_t        frame_first, frame_count;   // the frames this binning launch covers (a group of the batch)
    uint32_t        tile_first, tile_count;     // and their tiles
    const uint32_t* frame_off;    // [F+1] device copy
    const uint32_t* tile_base;    // [F+1] first tile of each frame

    const P3* xyz;                // input records: packed Point3D when in_stride == 12, else byte records
    uint32_t  in_stride;          // bytes per input record
    uint32_t  in_off[3];          // byte offsets of the FLOAT32 x, y, z fields inside a record (multiples of 4)
    SensorTable sensors;          // fusion pre-step (n = 0: off)
    const uint32_t* sensor_off;   // [F][sensors.n + 1] frame-relative first point of every sensor's raw cloud
    uint8_t*  labels;
    float*    residual;           // may be null

    uint32_t* hist;               // [n_tiles][PB]: the tile's points of every class (bin_tile's counting pass), turned by
                                  // scan_offsets into the frame-relative first slot of that run in the patch-sorted arrays
    uint32_t* ftot;               // [F][PB] points of every class in the frame (bin_tile adds its tile's counts; zeroed per batch)
    uint32_t* patch_begin;        // [F][P+1], frame-relative
    PatchDesc* patches;           // [F*P]
    uint32_t* bucket_count;       // [kBuckets]
    uint32_t* bucket_list;        // [kBuckets][list_stride]
    uint32_t  list_stride;

    float *sx, *sy, *sz;          // patch-sorted SoA, input order inside a patch (written by bin_tile's scatter pass)
    uint32_t* sidx;               // original index inside the frame, per slot (never permuted)
    float *ax, *ay, *az;          // scratch for the stable partition
    unsigned long long* mbits;    // inlier masks as bit planes: three planes of `mplane` words, one word per chunk of 64 node slots
    uint32_t mplane;              // (patch p of the batch starts at word (begin >> 6) + p of every plane)

    rpw_frame_stats* stats;       // [F]
    rpw_node* nodes;              // trace (may be null)
    uint32_t* node_count;
    uint32_t* work;               // [0] work counter of patch_fit's chain-server form, [1] its error flag (zeroed with the bucket counts)
    uint32_t  nodes_cap;
    int       trace;
    int       bin_remap;          // bin_tile: XCD-contiguous tile order
};

void launch_bin_count(const DevParams& p, hipStream_t s);
void launch_bin_scatter(const DevParams& p, hipStream_t s);
void launch_scan(const DevParams& p, hipStream_t s);
void launch_fit(const DevParams& p, hipStream_t s);
void launch_frame_setup(uint32_t* offsets, uint32_t n, uint32_t tiles, uint32_t* ctrl, uint32_t ctrl_words, hipStream_t s);
void launch_frame_copy_out(const uint8_t* labels, uint8_t* h_labels, uint32_t n, const float* res, float* h_res,
                           const rpw_frame_stats* stats, rpw_frame_stats* h_stats, const uint32_t* fit_error, hipStream_t s);
void launch_loop_bench(const float* sx, const float* sy, const float* sz, unsigned long long* mw, uint32_t mplane, uint32_t* il, uint32_t n, uint32_t count,
                       int what, int reps, uint32_t blocks, float* out, unsigned long long* ticks, hipStream_t s);
void launch_result_record(const rpw_frame_stats* stats, int32_t* out, uint32_t n_frames, const uint32_t* fit_error, hipStream_t s);
void launch_debug_angles(const float* x, const float* y, float* a, uint32_t n, hipStream_t s);
void launch_debug_eig3(const float* cov9, float* normal3, uint32_t n, hipStream_t s);

} // namespace rpw
