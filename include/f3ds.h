/* f3ds.h -- C-ABI of libf3ds: MI355X-native supervoxel + hierarchical-merge segmenter.
 *
 * One call replaces, for one XYZRGBA frame, the whole hot path of the reference
 * (citations are into /root/reference):
 *
 *   frame prelude  (z<0 -> |z|)                         src/supervoxel_clustering.cpp:313-340
 *   pcl::SupervoxelClustering<PointXYZRGBA> sequence    src/supervoxel_clustering.cpp:348-367
 *   Clustering::set_initialstate / cluster(threshold)   src/clustering.cpp:605-612, 670-679
 *   Clustering::get_labeled_cloud / get_colored_cloud   src/clustering.cpp:631-663
 *
 * The reference has no FFI of its own (it is a C++ CLI); a maintainer who wants to keep
 * main() and swap the path would bind exactly these entry points (INTEGRATION.md shows the
 * replacement block for main()).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Threading: one f3ds_ctx per host thread / stream; calls on one ctx are serial.
 * Errors: negative int codes (f3ds_strerror); nothing throws across this boundary.
 * Memory: caller owns every in/out buffer; the ctx owns its device scratch (grow-only).
 */
#ifndef F3DS_H_
#define F3DS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F3DS_VERSION 120

/* label written for points that belong to no region (non-finite input point, or a voxel no
 * supervoxel ever claimed).  The reference never emits such points at all
 * (Clustering::get_labeled_cloud only walks owned voxels, src/clustering.cpp:640-663). */
#define F3DS_NO_LABEL 0xFFFFFFFFu

/* error codes */
#define F3DS_OK 0
#define F3DS_ERR_ARG (-1)          /* null pointer / bad parameter value                        */
#define F3DS_ERR_NO_DEVICE (-2)    /* HIP runtime found no usable GPU                           */
#define F3DS_ERR_HIP (-3)          /* a HIP call failed (f3ds_last_hip_error has the text)      */
#define F3DS_ERR_DEPTH (-4)        /* voxel grid needs more than 21 octree levels               */
#define F3DS_ERR_LOGIC (-5)        /* call order (e.g. recluster before segment):
                                      std::logic_error in src/clustering.cpp:671-673            */
#define F3DS_ERR_RANGE (-6)        /* lambda outside [0,1] / bins < 0:
                                      std::invalid_argument in src/clustering.cpp:578,593       */
#define F3DS_ERR_UNSUPPORTED (-7)  /* degenerate input the device path refuses (documented)     */
#define F3DS_ERR_IO (-8)           /* PCD file could not be read / written                      */
#define F3DS_ERR_EQ_BIN (-9)       /* --EQ with delta_g == 1.0: map::at throws in the reference
                                      (src/clustering.cpp:371-372, t_g has no clamp)            */
#define F3DS_ERR_CAPACITY (-10)    /* output buffer too small                                   */
#define F3DS_ERR_BUSY (-11)        /* frame pipeline: every slot in flight / oldest frame not done yet */
#define F3DS_ERR_EMPTY (-12)       /* frame pipeline: nothing submitted that has not been taken */
#define F3DS_ERR_OUT_OF_RANGE (-13) /* std::out_of_range in the reference: an adjacency names a
                                      supervoxel label that is not in the set (map::at,
                                      src/clustering.cpp:228-229); all_thresh bounds outside
                                      [0,1] (:694-698)                                           */

/* enums mirror include/supervoxel_clustering/clustering.h:58-68 */
enum { F3DS_LAB_CIEDE00 = 0, F3DS_RGB_EUCL = 1 };
enum { F3DS_NORMALS_DIFF = 0, F3DS_CONVEX_NORMALS_DIFF = 1 };
enum { F3DS_MANUAL_LAMBDA = 0, F3DS_ADAPTIVE_LAMBDA = 1, F3DS_EQUALIZATION = 2 };

/* CLI flags of the reference (src/supervoxel_clustering.cpp:187-298) as a plain struct */
typedef struct f3ds_params {
    float voxel_res;          /* -v   default 0.008                                      */
    float seed_res;           /* -s   default 0.08                                       */
    float w_color;            /* -c   default 0.2                                        */
    float w_spatial;          /* -z   default 0.4                                        */
    float w_normal;           /* -n   default 1.0                                        */
    int32_t use_transform;    /* !--NT  (setUseSingleCameraTransform, :349)              */
    int32_t color_metric;     /* --RGB -> F3DS_RGB_EUCL, else F3DS_LAB_CIEDE00           */
    int32_t geom_metric;      /* --CVX -> F3DS_CONVEX_NORMALS_DIFF                       */
    int32_t merging;          /* --ML / --AL / --EQ                                      */
    float lambda;             /* --ML value; used only under F3DS_MANUAL_LAMBDA          */
    int32_t bins;             /* --EQ value; used only under F3DS_EQUALIZATION           */
    float threshold;          /* -t                                                      */
    int32_t leaf_order;       /* 0: octree leaves ascending (PCL >= 1.9), 1: descending  */
    int32_t fold_negative_z;  /* main()'s z<0 -> |z| (:317-321); the CLI always sets 1   */
} f3ds_params;

typedef struct f3ds_result {
    uint64_t n_points;            /* N                                                   */
    uint64_t n_finite;            /* points with finite x,y,z                            */
    uint32_t n_voxels;            /* V  occupied voxels                                  */
    uint32_t octree_depth;        /* levels of the voxel grid cube                       */
    uint32_t n_seed_cells;        /* occupied seed-resolution cells before filtering     */
    uint32_t n_seeds;             /* seeds kept = supervoxel helpers created             */
    uint32_t n_supervoxels;       /* S  non-empty supervoxels after the sweeps           */
    uint32_t n_edges;             /* E  undirected supervoxel adjacencies                */
    uint32_t n_merges;            /* merges performed below the threshold                */
    uint32_t n_regions;           /* K  regions left                                     */
    uint32_t sweeps;              /* max_depth-1 label-propagation sweeps                */
    float lambda;                 /* lambda actually used (Clustering::get_lambda)       */
    float ms_total;               /* wall time of the call, host clock                   */
    float ms_stage[8];            /* device time per stage (HIP events): 0 voxelise,
                                     1 neighbours+normals, 2 seeds, 3 sweeps, 4 supervoxel
                                     summaries+edges, 5 merge, 6 labels; 7 = the voxel-normal
                                     kernel's launch alone (part of stage 1): HIP event pair
                                     around that dispatch on the call's stream          */
} f3ds_result;

typedef struct f3ds_ctx f3ds_ctx;

void f3ds_default_params(f3ds_params* p);
int f3ds_version(void);
/* "f3ds 1.2.0 src:<16 hex digits>": the digits are the SHA-256 prefix of the sources the library was built from
 * (csrc/Makefile writes it at build time); tests and bench.py compare it with the sources on disk so that a stale
 * prebuilt libf3ds.so fails loudly.  A library built with the timing experiments compiled in ends in " +whatif"; a
 * process whose environment opens the development switches (below) gets " +dev" appended. */
const char* f3ds_version_string(void);
/* 1 when F3DS_DEV is set (to anything but "0") in the environment: only then does the library read its development
 * switches (F3DS_MERGE_NW, F3DS_VOX_TILES, ... -- kernel layouts and test hooks, none changes results).
 * Without it they are ignored, whatever the environment holds. */
int f3ds_dev_mode(void);
const char* f3ds_strerror(int code);
const char* f3ds_last_hip_error(void);

/* number of visible GPUs (0 when there is none); never initialises a device */
int f3ds_device_count(void);

/* create a context on `device` with its own stream.  Fails with F3DS_ERR_NO_DEVICE when the
 * HIP runtime has no GPU: there is no CPU fallback in this library. */
int f3ds_create(int device, f3ds_ctx** out);
void f3ds_destroy(f3ds_ctx* ctx);

/* run on a caller-provided hipStream_t (e.g. a torch stream) instead of the ctx's own */
int f3ds_set_stream(f3ds_ctx* ctx, void* hip_stream);

/* Segment one frame.  `points` is N records {float x,y,z; uint32 rgba} (16 B, rgba packed
 * a<<24|r<<16|g<<8|b as PCL does).  points_on_device / labels_on_device say where the caller's
 * buffers live.  point_labels receives N region ids (0..K-1 in the order of
 * Clustering::get_labeled_cloud, F3DS_NO_LABEL for points outside every region).
 * Synchronous: returns after the labels are in the caller's buffer. */
int f3ds_segment(f3ds_ctx* ctx, const void* points, size_t n, int points_on_device,
                 const f3ds_params* params, uint32_t* point_labels, int labels_on_device,
                 f3ds_result* result);

/* A batch of independent frames on one GPU, one context per frame (BASELINE.json config 5 puts 8
 * frames on each GPU).  Same result per frame as f3ds_segment; every kernel of the path is one
 * dispatch for all frames and all merge loops run as one dispatch.  points[i] / point_labels[i] /
 * counts[i] belong to ctxs[i]; results may be NULL.  Synchronous.  The batch runs on the stream set
 * with f3ds_set_stream on ctxs[0] if there is one, else on one of a few library-owned streams per
 * device (one per hardware queue), so that batch calls from several host threads run side by side.
 * Host buffers (points_on_device / labels_on_device == 0): the uploads and downloads of ALL calls on a device go
 * through one library-owned copy stream, one copy after the other -- the PCIe link moves 55 GB/s one way alone and ~16
 * each way when uploads and downloads of different calls overlap; the call's kernels wait for its uploads by event.
 * Pinned (hipHostMalloc) buffers are what makes these copies asynchronous; pageable ones work and are slower. */
int f3ds_segment_batch(f3ds_ctx** ctxs, int nctx, const void* const* points, const size_t* counts,
                       int points_on_device, const f3ds_params* params, uint32_t* const* point_labels,
                       int labels_on_device, f3ds_result* results);

/* Clustering::cluster(threshold) again on the supervoxels of the last f3ds_segment call, with
 * possibly different metric / merging settings (src/clustering.cpp:670-679).  Only the merge
 * fields of `params` are read. */
int f3ds_recluster(f3ds_ctx* ctx, const f3ds_params* params, uint32_t* point_labels,
                   int labels_on_device, f3ds_result* result);

/* Clustering::get_labeled_cloud / get_colored_cloud of the last call: one record per owned
 * voxel, regions in ascending label order, voxels in leaf order inside each supervoxel,
 * supervoxels in merge-concatenation order (src/clustering.cpp:640-663, 793-812).
 * Any of xyz / label / rgba may be NULL.  rgba = Glasbey[label % 256]. */
int f3ds_get_voxel_cloud(f3ds_ctx* ctx, float* xyz, uint32_t* label, uint32_t* rgba, size_t cap,
                         size_t* n_out);

/* parity hooks: copy an intermediate array of the last call to host memory */
enum {
    F3DS_DBG_GRID = 0,            /* 5 x f64: min_x, min_y, min_z, resolution, depth            */
    F3DS_DBG_VOXEL_KEYS = 1,      /* V x 3 u32, leaf order                                      */
    F3DS_DBG_VOXEL_COUNT = 2,     /* V u32 points per voxel                                     */
    F3DS_DBG_VOXEL_XYZ = 3,       /* V x 3 f32 centroid                                         */
    F3DS_DBG_VOXEL_RGB = 4,       /* V x 3 f32 mean colour                                      */
    F3DS_DBG_VOXEL_NORMAL = 5,    /* V x 4 f32 (w = 0)                                          */
    F3DS_DBG_VOXEL_NEIGHBORS = 6, /* V x 27 i32, slot = (dx+1)*9+(dy+1)*3+(dz+1), -1 = none     */
    F3DS_DBG_POINT_VOXEL = 7,     /* N i32 voxel of each input point, -1 = non-finite           */
    F3DS_DBG_SEED_ORIG = 8,       /* n_seed_cells i32: nearest voxel per occupied seed cell     */
    F3DS_DBG_SEED_KEPT = 9,       /* n_seeds i32: voxels that became helpers (label = index+1)  */
    F3DS_DBG_VOXEL_SVLABEL = 10,  /* V u32 supervoxel label after the sweeps, 0 = none          */
    F3DS_DBG_VOXEL_DIST = 11,     /* V f32 VoxelData::distance_ after the sweeps                */
    F3DS_DBG_SV_LABELS = 12,      /* S u32 labels of non-empty supervoxels, ascending           */
    F3DS_DBG_SV_CENTROID = 13,    /* S x 10 f32: xyz, rgb, normal4 of each (same order)         */
    F3DS_DBG_EDGES = 14,          /* E x 2 u32 (a<b), sorted                                    */
    F3DS_DBG_EDGE_DELTAS = 15,    /* E x 2 f32 (delta_c, delta_g)                               */
    F3DS_DBG_EDGE_WEIGHTS = 16,   /* E f32 initial weights                                      */
    F3DS_DBG_MERGES = 17,         /* n_merges x 3 u32: a, b, weight bits                        */
    F3DS_DBG_VOXEL_REGION = 18,   /* V u32 final region id per voxel (F3DS_NO_LABEL = none)     */
    F3DS_DBG_SV_REGION = 19,      /* S u32 surviving label each supervoxel ended in             */
    F3DS_DBG_TILE_LIST_LEN = 21,  /* ceil(V / 128) u32: length of each 128-voxel tile's one-ring list (the LDS tiles of the normals and
                                     the sweeps); 0xFFFFFFFF = the tile did not fit the tables and took the global-memory path.  Diagnostics */
    F3DS_DBG_SWEEP_STATS = 22,    /* 4 u32: label-propagation sweeps of the last run that evaluated every voxel from their start / only the tiles marked
                                     dirty (incremental R rounds) / started incremental and fell back to the chain walker because the last R round
                                     still changed a word / were skipped because the sweep before them had changed nothing (a fixed point: the
                                     remaining iterations of expandSupervoxels are no-ops).  Diagnostics and tests (all kinds end in the same bits) */
    F3DS_DBG_STAGE0_PATH = 23,    /* 1 u32: how the last frame was voxelised -- 1 = the tile path (the points stay where they are, only per-tile voxel
                                     descriptors are sorted: frames of a batch), 0 = the sort path (every point's key through a radix sort: lone frames,
                                     unorganised clouds, very dense voxels).  Diagnostics and tests: the two paths give the same bits              */
    F3DS_DBG_MERGE_LAYOUT = 20    /* 2 u32: which merge kernel the last cluster stage ran -- waves per frame (4, 8; 0 = d_merge,
                                     everything in global memory) and where it kept the per-edge arrays (2 = order keys + endpoints in
                                     LDS, 0 = in global memory).  Diagnostics (bench.py names the kernel it timed): results do not depend on it */
};
int f3ds_get_debug(f3ds_ctx* ctx, int what, void* dst, size_t cap_bytes, size_t* bytes_out);
/* Diagnostics, host arithmetic only (no device needed): the LDS carve-up of the merge kernel d_merge_il_t<waves, keys_in_lds> for a frame with n_edges
 * adjacencies -- the same function the launch and the kernel use (csrc/f3ds_kernels.inc, merge_il_offsets).  out[0] = dynamic LDS bytes, out[1] = voxel rows the
 * speculative second merge of an epoch may absorb (128; 256, 512 or 1024 in the 8-wave layout where LDS has room), out[2] = edge slots, out[3] = 1 if the layout fits a
 * compute unit's LDS (else the cluster stage takes the next layout, at last d_merge).  waves: 4 or 8; keys_in_lds: 2 or 0.  Returns F3DS_ERR_ARG otherwise. */
int f3ds_merge_layout_info(uint32_t n_edges, int waves, int keys_in_lds, uint32_t out[4]);

/* ---- the VCCS result itself: what main() reads back from pcl::SupervoxelClustering after extract()
 * (src/supervoxel_clustering.cpp:359-367).  All valid after f3ds_segment; call with NULL outputs for the count. */

/* getVoxelCentroidCloud (:359) + getLabeledVoxelCloud: one entry per voxel in leaf order: centroid, mean colour
 * as 0x00RRGGBB (VoxelData::getPoint truncation), supervoxel label (0 = none). */
int f3ds_get_voxel_centroid_cloud(f3ds_ctx* ctx, float* xyz, uint32_t* rgba, uint32_t* sv_label, size_t cap, size_t* n_out);

/* the supervoxel_clusters map + makeSupervoxelNormalCloud (:356,360): per non-empty supervoxel, ascending label:
 * label, centroid xyz, mean rgb (float), unit normal, number of voxels. */
int f3ds_get_supervoxels(f3ds_ctx* ctx, uint32_t* label, float* xyz, float* rgb, float* normal, uint32_t* n_voxels,
                         size_t cap, size_t* n_out);

/* getSupervoxelAdjacency (:365) after clear_adjacency: pairs (a < b) of adjacent supervoxel labels, sorted. */
int f3ds_get_supervoxel_adjacency(f3ds_ctx* ctx, uint32_t* pairs, size_t cap_pairs, size_t* n_out);

/* get_currentstate().second after cluster() (:444): adjacency of the merged regions as sorted pairs (a < b) of the surviving
 * supervoxel labels -- the graph visualize() draws between the centroids of supervoxel_clusters.at(label) (:636-672). */
int f3ds_get_region_adjacency(f3ds_ctx* ctx, uint32_t* pairs, size_t cap_pairs, size_t* n_out);

/* ---- the merge half on its own: Clustering on supervoxels the CALLER supplies ------------------------------------------
 * Clustering::set_initialstate(ClusteringT segm, AdjacencyMapT adj) takes "the output of some supervoxel algorithm"
 * (include/supervoxel_clustering/clustering.h:142, src/clustering.cpp:600-612; main() passes PCL's own supervoxel_clusters
 * and getSupervoxelAdjacency result, src/supervoxel_clustering.cpp:365,424).  This entry is that call + cluster(threshold)
 * (src/clustering.cpp:670-679) on plain arrays: a maintainer with PCL runs real PCL supervoxels through the GPU merge loop.
 *
 * One f3ds_supervoxel_set row per map entry (pcl::Supervoxel<PointXYZRGBA>): label = the map key (distinct, any u32, any
 * order: the ascending-key iteration of std::map is rebuilt here); voxels_ = voxel_xyz / voxel_rgba[voxel_offset[i] ..
 * voxel_offset[i+1]) in the cloud's own order (rgba packed as PCL does, a<<24|r<<16|g<<8|b; only r g b are read:
 * ColorUtilities::mean_color, src/color_utilities.cpp:117-142); centroid_ (x y z) and normal_ (normal_x/y/z).  Every
 * supervoxel must hold at least one voxel (PCL emits no empty ones).
 * adjacency_pairs = the multimap in ITERATION order, n_pairs (first, second) rows; rows with first > second are dropped
 * (clear_adjacency, :476-486), the rest become the initial weight-map entries in that order (adj2weight: all keys -1, so
 * insertion order; ties between equal weights later resolve in this order).  Rows are stably sorted by `first` in case the
 * caller did not iterate a multimap.  F3DS_ERR_OUT_OF_RANGE for an endpoint that is no row of the set (map::at throws
 * std::out_of_range, :228-229); F3DS_ERR_ARG for a pair listed twice or a self-adjacency (the reference dereferences an
 * erased supervoxel at the first merge that meets one).
 * Outputs (either may be NULL): region_of_sv[i] = label of the surviving supervoxel row i ended in (the key it has in
 * get_currentstate().first); voxel_labels[v] = get_labeled_cloud() id (0..K-1, regions in ascending key) of input voxel v.
 * Afterwards the context answers f3ds_recluster (other metric / threshold on the same supervoxels), f3ds_get_voxel_cloud,
 * f3ds_get_regions, f3ds_get_region_voxels, f3ds_get_region_adjacency, f3ds_get_supervoxel_adjacency and the merge-side
 * F3DS_DBG_* arrays (labels are the caller's); the VCCS-side accessors return F3DS_ERR_LOGIC. */
typedef struct f3ds_supervoxel_set {
    uint32_t n_supervoxels;
    const uint32_t* label;          /* n_supervoxels                                      */
    const uint32_t* voxel_offset;   /* n_supervoxels + 1, ascending, voxel_offset[0] = 0  */
    const float* voxel_xyz;         /* 3 per voxel                                        */
    const uint32_t* voxel_rgba;     /* 1 per voxel                                        */
    const float* centroid_xyz;      /* 3 per supervoxel                                   */
    const float* normal;            /* 3 per supervoxel                                   */
} f3ds_supervoxel_set;
int f3ds_cluster_supervoxels(f3ds_ctx* ctx, const f3ds_supervoxel_set* sv, const uint32_t* adjacency_pairs, size_t n_pairs,
                             const f3ds_params* params, uint32_t* region_of_sv, uint32_t* voxel_labels, f3ds_result* result);

/* get_currentstate().first after cluster() (src/clustering.cpp:619-624; read at src/supervoxel_clustering.cpp:443-449): the
 * merged regions in ascending key.  Per region: its key (label of the surviving supervoxel), number of voxels, centroid_
 * (computeCentroid of the concatenated voxels_, :412-414), normal_ (:416-425) and the mean colour mean_color() gives.  Any
 * output may be NULL.  Valid after f3ds_segment / f3ds_recluster / f3ds_cluster_supervoxels. */
int f3ds_get_regions(f3ds_ctx* ctx, uint32_t* label, uint32_t* n_voxels, float* centroid_xyz, float* normal, float* mean_rgb,
                     size_t cap, size_t* n_out);
/* the regions' voxels_ clouds, concatenated in the order of f3ds_get_regions (n_voxels[] splits them), each in the
 * concatenation order merge() builds (voxels_a ++ voxels_b, :408): position, colour (0x00RRGGBB, the voxel's own, not the
 * Glasbey colour of f3ds_get_voxel_cloud) and the voxel's index -- leaf ordinal after f3ds_segment (the row of
 * f3ds_get_voxel_centroid_cloud), input voxel index after f3ds_cluster_supervoxels -- with which the caller gathers any other
 * per-voxel attribute (Supervoxel::normals_, :409).  Any output may be NULL. */
int f3ds_get_region_voxels(f3ds_ctx* ctx, float* xyz, uint32_t* rgba, uint32_t* voxel_index, size_t cap, size_t* n_out);

/* ---- evaluation against ground truth and automatic threshold (the path run when -t is omitted) ----
 * Mirrors Testing::eval_performance (src/testing.cpp:239-406) and Clustering::all_thresh / best_thresh
 * (src/clustering.cpp:691-774) on the frame of the last f3ds_segment call.  truth_point_labels holds
 * one ground-truth label per input point (the PCD `label` field).  The truth cloud is built like
 * main() does (src/supervoxel_clustering.cpp:387-400): points coloured by label -> voxel mean colour ->
 * one truth label per distinct voxel colour, in first-appearance order. */
typedef struct f3ds_performance { float voi, precision, recall, fscore, wov, fpr, fnr; } f3ds_performance;

/* scores of the current segmentation (Testing(get_labeled_cloud(), truth).eval_performance()) */
int f3ds_evaluate(f3ds_ctx* ctx, const uint32_t* truth_point_labels, f3ds_performance* out);

/* all_thresh(start, end, step) + best_thresh: clusters at every threshold start, start+step, ... <= end
 * (float accumulation as in the reference), scores each, returns the best by F-score and leaves the
 * context clustered at that threshold (point_labels receives its labels; may be NULL).
 * thresholds / scores (capacity cap) receive the sweep; n_out its length. */
int f3ds_auto_threshold(f3ds_ctx* ctx, const f3ds_params* params, const uint32_t* truth_point_labels,
                        float start, float end, float step, float* thresholds, f3ds_performance* scores,
                        size_t cap, size_t* n_out, float* best_threshold, f3ds_performance* best_score,
                        uint32_t* point_labels, int labels_on_device, f3ds_result* result);

/* SupervoxelClustering::refineSupervoxels(num_itr, refined_supervoxel_clusters) (src/supervoxel_clustering.cpp:369-375) on
 * the supervoxels of the last f3ds_segment call: num_itr times { normals again from the owned two-ring, reseed every
 * supervoxel at the voxel nearest to its centroid, expand again }.  As in the reference, the refined supervoxels feed
 * nothing downstream (the clustering keeps the unrefined ones, :424): they live beside the frame's state and are read
 * with the two getters below; every other call keeps describing the unrefined supervoxels. */
int f3ds_refine_supervoxels(f3ds_ctx* ctx, int num_itr);
/* per voxel in leaf order (the order of f3ds_get_voxel_centroid_cloud): refined supervoxel label (0 = none) --
 * getLabeledVoxelCloud / the per-voxel view of refined_full_labeled_cloud (:375) -- and refined normal (3 floats) */
int f3ds_get_refined_voxels(f3ds_ctx* ctx, uint32_t* sv_label, float* normal, size_t cap, size_t* n_out);
/* refined_supervoxel_clusters / makeSupervoxelNormalCloud(refined) (:373-374): layout of f3ds_get_supervoxels */
int f3ds_get_refined_supervoxels(f3ds_ctx* ctx, uint32_t* label, float* xyz, float* rgb, float* normal, uint32_t* n_voxels,
                                 size_t cap, size_t* n_out);

/* ---- frame pipeline (row N4: the streaming surface the reference's ROS launch file points at,
 * launch/supervoxel_clustering.launch:3-6, README.md:72) ----------------------------------------
 * Frames go in from host memory one at a time and come out in submission order.  Up to `depth`
 * frames are in flight: each has a context and pinned staging for its points and labels, `groups`
 * host threads pick up whatever is queued -- one frame when the producer is slow (latency of a lone
 * frame), a run of consecutive frames with equal parameters as one f3ds_segment_batch when frames
 * arrive faster than they finish (throughput of the batched path) -- so H2D of one group overlaps
 * the kernels and the merge loop of the others.  Results per frame are those of f3ds_segment.
 * submit/next may be called from different threads (one producer, one consumer). */
typedef struct f3ds_stream f3ds_stream;
int f3ds_stream_create(int device, int depth, int groups, f3ds_stream** out);
void f3ds_stream_destroy(f3ds_stream* s);
/* pinned input buffer of the slot the next submit will use, sized for n points; filling it in place
 * (e.g. f3ds_pcd_read straight into it) saves submit's copy.  F3DS_ERR_BUSY when depth frames are in flight. */
int f3ds_stream_buffer(f3ds_stream* s, size_t n, void** points16);
/* queue a frame (host memory; copied unless it is the pointer f3ds_stream_buffer gave).  Never blocks:
 * F3DS_ERR_BUSY when depth frames are in flight -- take one with f3ds_stream_next first. */
int f3ds_stream_submit(f3ds_stream* s, const void* points, size_t n, const f3ds_params* params, uint64_t tag);
/* oldest frame not taken yet: its labels (n_out of them) into point_labels, its tag and result.  wait != 0 blocks
 * until it is done, else F3DS_ERR_BUSY while it is running; F3DS_ERR_EMPTY when nothing is in flight;
 * F3DS_ERR_CAPACITY (frame stays) when cap < its point count; n_out and tag are set in all three cases but EMPTY.
 * Otherwise the frame leaves the pipeline and the call returns the frame's own status. */
int f3ds_stream_next(f3ds_stream* s, uint32_t* point_labels, size_t cap, size_t* n_out, uint64_t* tag, f3ds_result* result, int wait);
/* the oldest frame's labels where they are (the slot's pinned buffer) instead of a copy: valid until the frame is
 * taken with f3ds_stream_next(s, NULL, 0, ...).  Same waiting rule and return values as f3ds_stream_next. */
int f3ds_stream_peek(f3ds_stream* s, const uint32_t** point_labels, size_t* n_out, uint64_t* tag, f3ds_result* result, int wait);
/* frames submitted and not taken */
int f3ds_stream_pending(f3ds_stream* s);

/* ---- multi-GPU batch driver, one process (BASELINE.json config 5: a batch of independent frames sharded over the
 * GPUs of a node, label output gathered on one GPU).  The reference has no counterpart (single-threaded CLI,
 * CMakeLists.txt:5).  Frame i runs on devices[i mod n_devices], one host thread per GPU; the per-point labels of all
 * frames then go to devices[0] in ONE grouped RCCL send/recv exchange (the ragged form of ncclGather, each peer over its
 * own xGMI link) and from there to the caller's host buffers.  librccl is loaded at run time; with one device no RCCL
 * call is made (unless F3DS_MULTI_FORCE_RCCL is set: development).  Results per frame are those of f3ds_segment.
 * F3DS_MULTI_LOGICAL=1 (tests on 1-GPU boxes): `devices` may name one GPU several times -- every entry is a device of its own
 * to the driver (worker thread, contexts, label blocks), the exchange a device-to-device copy in place of ncclSend / ncclRecv. */
typedef struct f3ds_multi f3ds_multi;
/* devices == NULL: devices 0..n_devices-1.  max_frames_per_device bounds one f3ds_multi_segment call. */
int f3ds_multi_create(const int* devices, int n_devices, int max_frames_per_device, f3ds_multi** out);
void f3ds_multi_destroy(f3ds_multi* m);
int f3ds_multi_devices(const f3ds_multi* m);
/* index into `devices` of the GPU frame `frame` of a call runs on */
int f3ds_multi_device_of_frame(const f3ds_multi* m, int frame);
/* points[i]: counts[i] records of 16 B in host memory; point_labels[i]: host buffer of counts[i] uint32 (may be NULL);
 * results may be NULL.  F3DS_ERR_CAPACITY when n_frames > n_devices * max_frames_per_device. */
int f3ds_multi_segment(f3ds_multi* m, const void* const* points, const size_t* counts, int n_frames, const f3ds_params* params,
                       uint32_t* const* point_labels, f3ds_result* results);
/* Pipelined form: f3ds_multi_submit queues a batch (same arguments; the pointer arrays are copied; the point buffers, the label
 * buffers AND the `results` array are written / read asynchronously by the driver's threads and must stay valid until the batch
 * is collected) and returns a ticket; the GPUs compute batch k+1 while batch k's labels are gathered on
 * devices[0] and copied to the host buffers.  At most two batches are in flight: F3DS_ERR_BUSY when the batch submitted two
 * calls ago has not been collected yet.  f3ds_multi_collect waits for a batch and returns its status (once per ticket).
 * f3ds_multi_segment == submit + collect. */
int f3ds_multi_submit(f3ds_multi* m, const void* const* points, const size_t* counts, int n_frames, const f3ds_params* params,
                      uint32_t* const* point_labels, f3ds_result* results, int* ticket);
int f3ds_multi_collect(f3ds_multi* m, int ticket);
/* allocate the label blocks for batches of max_frames_per_device frames of up to max_points_per_frame points now (otherwise they
 * grow inside the first batches); F3DS_ERR_BUSY while a batch is in flight */
int f3ds_multi_reserve(f3ds_multi* m, size_t max_points_per_frame);
/* the gathered label block on devices[0] of the batch gathered last (device pointer): device d's frames, in frame order, start
 * at the sum of the point counts of the devices before d; valid until the batch after the next one is submitted */
const uint32_t* f3ds_multi_gathered_labels(const f3ds_multi* m);
/* the gathered block of a given batch (ticket of f3ds_multi_submit): NULL unless that batch has been gathered without error and its
 * slot has not been handed to a later batch */
const uint32_t* f3ds_multi_gathered_labels_of(f3ds_multi* m, int ticket);
const char* f3ds_multi_last_error(void);

/* ---- host-side helpers either side of the path (no GPU needed) -------------------------- */

/* PCD v0.7 reader (ascii / binary / binary_compressed), replaces pcl::io::loadPCDFile at
 * src/supervoxel_clustering.cpp:313.  Fields x y z + rgb|rgba (+ optional label).  Call with
 * points == NULL to get the point count.  labels may be NULL; missing label field -> 0. */
int f3ds_pcd_read(const char* path, void* points16, uint32_t* labels, size_t cap, size_t* n_out,
                  uint32_t* width, uint32_t* height);
/* PCD writer: fields "x y z rgba" (+ "label" when labels != NULL); mode 0 ascii, 1 binary */
int f3ds_pcd_write(const char* path, const float* xyz, const uint32_t* rgba,
                   const uint32_t* labels, size_t n, int mode);

/* deterministic synthetic frames (SplitMix64): kind 0 = pinhole RGB-D room (width x height
 * pixels), kind 1 = fused multi-view room scene with width*height
 * samples (config 4).  nan_permille = invalid-depth pixels per thousand. */
int f3ds_synth_frame(int kind, uint64_t seed, uint32_t width, uint32_t height,
                     uint32_t nan_permille, void* points16);

/* Glasbey-style lookup used for the coloured cloud (256 entries, r<<16|g<<8|b) */
uint32_t f3ds_label_color(uint32_t label);

#ifdef __cplusplus
}
#endif
#endif /* F3DS_H_ */
