// f3ds_clustering.hpp -- header-only C++ surface over the C-ABI of include/f3ds.h, shaped like the classes a user of the
// reference already calls (citations into /root/reference):
//
//   f3ds::SupervoxelClustering  the nine-call pcl::SupervoxelClustering<PointXYZRGBA> sequence of main()
//                               (src/supervoxel_clustering.cpp:348-367): same method names, one object
//   f3ds::Clustering            class Clustering, include/supervoxel_clustering/clustering.h:116-211: same constructors,
//                               setters, getters, cluster(), all_thresh / best_thresh, get_labeled_cloud /
//                               get_colored_cloud; std::logic_error / std::invalid_argument where the reference throws
//                               them (src/clustering.cpp:574-597, 670-673, 693-700)
//
// set_initialstate has the reference's form -- a map of supervoxels of ANY algorithm + an adjacency multimap, copied by value
// (clustering.h:142, clustering.cpp:605-612; f3ds::Supervoxel stands for pcl::Supervoxel<PointXYZRGBA>) -- and a second form that
// takes the SupervoxelClustering object, whose extract() result is already device-resident.  Clouds come back as plain arrays
// (xyz triples + label / rgba), not pcl::PointCloud.  Nothing here touches HIP: link against libf3ds.so only.
#ifndef F3DS_CLUSTERING_HPP_
#define F3DS_CLUSTERING_HPP_

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "f3ds.h"

namespace f3ds {

enum ColorDistance { LAB_CIEDE00 = F3DS_LAB_CIEDE00, RGB_EUCL = F3DS_RGB_EUCL };                                  // clustering.h:62-64
enum GeometricDistance { NORMALS_DIFF = F3DS_NORMALS_DIFF, CONVEX_NORMALS_DIFF = F3DS_CONVEX_NORMALS_DIFF };      // :66-68
enum MergingCriterion { MANUAL_LAMBDA = F3DS_MANUAL_LAMBDA, ADAPTIVE_LAMBDA = F3DS_ADAPTIVE_LAMBDA, EQUALIZATION = F3DS_EQUALIZATION };   // :70-72

typedef f3ds_performance performanceSet;                       // include/supervoxel_clustering/testing.h:40-48

struct PointXYZRGBA { float x, y, z; uint32_t rgba; };         // the 16-byte record of the C-ABI (PCL packs rgba the same way)
struct LabeledCloud { std::vector<float> xyz; std::vector<uint32_t> label; };      // get_labeled_cloud(): one entry per owned voxel
struct ColoredCloud { std::vector<float> xyz; std::vector<uint32_t> rgba; };       // get_colored_cloud()
struct Supervoxels { std::vector<uint32_t> label, n_voxels; std::vector<float> xyz, rgb, normal; };   // the supervoxel_clusters map, ascending label
// pcl::Supervoxel<PointXYZRGBA> as Clustering reads it (clustering.cpp:108-129,405-425): voxels_, centroid_, normal_.  voxel_index is filled by
// get_currentstate(): position of each voxel in the initial state (leaf ordinal, or index into the concatenated input voxels_), with which a caller
// gathers per-voxel attributes such as normals_
struct Supervoxel { std::vector<PointXYZRGBA> voxels; float centroid[3] = {0, 0, 0}; float normal[3] = {0, 0, 0}; float mean_rgb[3] = {0, 0, 0}; std::vector<uint32_t> voxel_index; };
typedef std::map<uint32_t, Supervoxel> ClusteringT;                // clustering.h:75
typedef std::multimap<uint32_t, uint32_t> AdjacencyMapT;           // clustering.h:76

inline void check(int rc, const char* what) {
    if (rc == F3DS_OK) return;
    const std::string msg = std::string(what) + ": " + f3ds_strerror(rc) + (rc == F3DS_ERR_HIP ? std::string(" [") + f3ds_last_hip_error() + "]" : std::string());
    if (rc == F3DS_ERR_LOGIC) throw std::logic_error(msg);
    if (rc == F3DS_ERR_OUT_OF_RANGE) throw std::out_of_range(msg);      // map::at / all_thresh bounds (clustering.cpp:228-229, 694-698)
    if (rc == F3DS_ERR_RANGE || rc == F3DS_ERR_ARG) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

class SupervoxelClustering {
    f3ds_ctx* ctx_ = nullptr;
    f3ds_params prm_;
    const void* cloud_ = nullptr;
    size_t n_ = 0;
    bool extracted_ = false;
    friend class Clustering;

public:
    SupervoxelClustering(float voxel_resolution, float seed_resolution, int device = 0) {      // :348
        f3ds_default_params(&prm_);
        prm_.voxel_res = voxel_resolution; prm_.seed_res = seed_resolution;
        check(f3ds_create(device, &ctx_), "f3ds_create");
    }
    ~SupervoxelClustering() { f3ds_destroy(ctx_); }
    SupervoxelClustering(const SupervoxelClustering&) = delete;
    SupervoxelClustering& operator=(const SupervoxelClustering&) = delete;

    void setUseSingleCameraTransform(bool v) { prm_.use_transform = v ? 1 : 0; }             // :349
    void setInputCloud(const PointXYZRGBA* points, size_t n) { cloud_ = points; n_ = n; extracted_ = false; }      // :351 (not copied: keep it alive until extract())
    void setColorImportance(float v) { prm_.w_color = v; }                                  // :352
    void setSpatialImportance(float v) { prm_.w_spatial = v; }                              // :353
    void setNormalImportance(float v) { prm_.w_normal = v; }                                // :354
    void setFoldNegativeZ(bool v) { prm_.fold_negative_z = v ? 1 : 0; }                     // main()'s z<0 -> |z| (:317-321), applied on the device
    void setLeafOrder(int order) { prm_.leaf_order = order; }                               // 0: PCL >= 1.9, 1: PCL 1.8

    // extract(supervoxel_clusters) (:356).  Runs VCCS (and, because the C-ABI has one entry point for the frame, a first
    // clustering at threshold 0, i.e. no merge); Clustering::cluster() then re-clusters the same supervoxels.
    Supervoxels extract() {
        if (!cloud_ && n_) throw std::logic_error("setInputCloud first");
        f3ds_params p = prm_; p.threshold = 0.0f;
        check(f3ds_segment(ctx_, cloud_, n_, 0, &p, nullptr, 0, nullptr), "f3ds_segment");
        extracted_ = true;
        return supervoxels();
    }
    Supervoxels supervoxels() const {
        size_t k = 0;
        check(f3ds_get_supervoxels(ctx_, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &k), "f3ds_get_supervoxels");
        Supervoxels s; s.label.resize(k); s.n_voxels.resize(k); s.xyz.resize(3 * k); s.rgb.resize(3 * k); s.normal.resize(3 * k);
        if (k) check(f3ds_get_supervoxels(ctx_, s.label.data(), s.xyz.data(), s.rgb.data(), s.normal.data(), s.n_voxels.data(), k, &k), "f3ds_get_supervoxels");
        return s;
    }
    ColoredCloud getVoxelCentroidCloud() const {                                             // :359
        size_t v = 0;
        check(f3ds_get_voxel_centroid_cloud(ctx_, nullptr, nullptr, nullptr, 0, &v), "f3ds_get_voxel_centroid_cloud");
        ColoredCloud c; c.xyz.resize(3 * v); c.rgba.resize(v);
        if (v) check(f3ds_get_voxel_centroid_cloud(ctx_, c.xyz.data(), c.rgba.data(), nullptr, v, &v), "f3ds_get_voxel_centroid_cloud");
        return c;
    }
    LabeledCloud getLabeledVoxelCloud() const {
        size_t v = 0;
        check(f3ds_get_voxel_centroid_cloud(ctx_, nullptr, nullptr, nullptr, 0, &v), "f3ds_get_voxel_centroid_cloud");
        LabeledCloud c; c.xyz.resize(3 * v); c.label.resize(v);
        if (v) check(f3ds_get_voxel_centroid_cloud(ctx_, c.xyz.data(), nullptr, c.label.data(), v, &v), "f3ds_get_voxel_centroid_cloud");
        return c;
    }
    std::multimap<uint32_t, uint32_t> getSupervoxelAdjacency() const {                      // :365, after clear_adjacency (a < b)
        size_t e = 0;
        check(f3ds_get_supervoxel_adjacency(ctx_, nullptr, 0, &e), "f3ds_get_supervoxel_adjacency");
        std::vector<uint32_t> pairs(2 * e);
        if (e) check(f3ds_get_supervoxel_adjacency(ctx_, pairs.data(), e, &e), "f3ds_get_supervoxel_adjacency");
        std::multimap<uint32_t, uint32_t> m;
        for (size_t i = 0; i < e; ++i) m.insert({pairs[2 * i], pairs[2 * i + 1]});
        return m;
    }
    void refineSupervoxels(int num_itr) { check(f3ds_refine_supervoxels(ctx_, num_itr), "f3ds_refine_supervoxels"); }     // :371
    f3ds_ctx* context() const { return ctx_; }
    size_t size() const { return n_; }
    const f3ds_params& params() const { return prm_; }
};

class Clustering {
    ColorDistance delta_c_type;
    GeometricDistance delta_g_type;
    MergingCriterion merging_type;
    float lambda;
    short bins_num;
    SupervoxelClustering* super_ = nullptr;
    std::vector<uint32_t> point_labels_;
    f3ds_result res_{};
    // set_initialstate(segm, adj): the caller's supervoxels as the arrays f3ds_cluster_supervoxels takes (copied: value semantics as in the reference)
    struct UserState {
        f3ds_ctx* ctx = nullptr; bool uploaded = false;
        std::vector<uint32_t> label, offset, rgba, pairs, region_of_sv; std::vector<float> xyz, centroid, normal;
    } user_;
    bool have_user_ = false;

    f3ds_ctx* ctx() const { return have_user_ ? user_.ctx : super_->ctx_; }
    size_t n_labels() const { return have_user_ ? user_.rgba.size() : super_->n_; }
    f3ds_params params(float threshold) const {
        f3ds_params p; if (have_user_) f3ds_default_params(&p); else p = super_->prm_;
        p.color_metric = delta_c_type; p.geom_metric = delta_g_type; p.merging = merging_type;
        p.lambda = merging_type == MANUAL_LAMBDA ? lambda : 0.0f;
        p.bins = merging_type == EQUALIZATION ? bins_num : 0;
        p.threshold = threshold;
        return p;
    }
    void need_state(const char* who) const {
        if (have_user_) return;
        if (!super_ || !super_->extracted_) throw std::logic_error(std::string("Cannot call '") + who + "' before setting an initial state with 'set_initialstate'");
    }

public:
    Clustering() : Clustering(LAB_CIEDE00, NORMALS_DIFF, ADAPTIVE_LAMBDA) {}                                     // clustering.cpp:533-539
    Clustering(ColorDistance c, GeometricDistance g, MergingCriterion m) : delta_c_type(c), delta_g_type(g) { set_merging(m); }
    ~Clustering() { f3ds_destroy(user_.ctx); }
    Clustering(const Clustering&) = delete;
    Clustering& operator=(const Clustering&) = delete;

    void set_delta_c(ColorDistance d) { delta_c_type = d; }
    void set_delta_g(GeometricDistance d) { delta_g_type = d; }
    void set_merging(MergingCriterion m) { merging_type = m; lambda = 0.5f; bins_num = 500; }                   // :562-567
    void set_lambda(float l) {                                                                                  // :574-582
        if (merging_type != MANUAL_LAMBDA) throw std::logic_error("Lambda can be set only if the merging criterion is set to MANUAL_LAMBDA");
        if (l < 0 || l > 1) throw std::invalid_argument("Argument outside range [0, 1]");
        lambda = l;
    }
    void set_bins_num(short b) {                                                                                // :589-597
        if (merging_type != EQUALIZATION) throw std::logic_error("Bins number can be set only if the merging criterion is set to EQUALIZATION");
        if (b < 0) throw std::invalid_argument("Argument lower than 0");
        bins_num = b;
    }
    // set_initialstate(segm, adj) (clustering.h:142, clustering.cpp:605-612): supervoxels of any algorithm + their adjacency, by value.
    // Errors surface at cluster(), where the reference meets them (init_weights, :212-251): std::out_of_range for an adjacency
    // naming a label that is not in segm (map::at), std::invalid_argument for what the reference leaves undefined (an adjacency
    // listed twice, a self-adjacency, an empty supervoxel).
    void set_initialstate(const ClusteringT& segm, const AdjacencyMapT& adj, int device = 0) {
        UserState u; u.ctx = user_.ctx; user_.ctx = nullptr;
        u.offset.push_back(0);
        for (const auto& kv : segm) {
            u.label.push_back(kv.first);
            for (const PointXYZRGBA& p : kv.second.voxels) { u.xyz.push_back(p.x); u.xyz.push_back(p.y); u.xyz.push_back(p.z); u.rgba.push_back(p.rgba); }
            u.offset.push_back((uint32_t)u.rgba.size());
            for (int a = 0; a < 3; ++a) { u.centroid.push_back(kv.second.centroid[a]); u.normal.push_back(kv.second.normal[a]); }
        }
        for (const auto& kv : adj) { u.pairs.push_back(kv.first); u.pairs.push_back(kv.second); }
        if (!u.ctx) check(f3ds_create(device, &u.ctx), "f3ds_create");
        user_ = std::move(u); have_user_ = true; super_ = nullptr; point_labels_.clear();
    }
    // the same on the device-resident result of `sv`.extract()
    void set_initialstate(SupervoxelClustering& sv) { super_ = &sv; have_user_ = false; point_labels_.clear(); }

    ColorDistance get_delta_c() const { return delta_c_type; }
    GeometricDistance get_delta_g() const { return delta_g_type; }
    MergingCriterion get_merging() const { return merging_type; }
    float get_lambda() const { return lambda; }
    short get_bins_num() const { return bins_num; }

    void cluster(float threshold) {                                                                             // :670-679
        need_state("cluster");
        const f3ds_params p = params(threshold);
        point_labels_.resize(n_labels());
        if (have_user_ && !user_.uploaded) {
            f3ds_supervoxel_set s;
            s.n_supervoxels = (uint32_t)user_.label.size(); s.label = user_.label.data(); s.voxel_offset = user_.offset.data(); s.voxel_xyz = user_.xyz.data();
            s.voxel_rgba = user_.rgba.data(); s.centroid_xyz = user_.centroid.data(); s.normal = user_.normal.data();
            user_.region_of_sv.resize(user_.label.size());
            check(f3ds_cluster_supervoxels(user_.ctx, &s, user_.pairs.data(), user_.pairs.size() / 2, &p, user_.region_of_sv.data(), point_labels_.data(), &res_),
                  "f3ds_cluster_supervoxels");
            user_.uploaded = !user_.label.empty();
        } else {
            check(f3ds_recluster(ctx(), &p, point_labels_.data(), 0, &res_), "f3ds_recluster");
            // a later cluster(t) moves the supervoxels to other regions: get_region_of_supervoxel() follows (F3DS_DBG_SV_REGION = the surviving
            // label per supervoxel in ascending label order, i.e. the row order of segm)
            if (have_user_ && !user_.label.empty()) {
                size_t nb = 0;
                user_.region_of_sv.assign(user_.label.size(), 0u);
                check(f3ds_get_debug(ctx(), F3DS_DBG_SV_REGION, user_.region_of_sv.data(), user_.region_of_sv.size() * sizeof(uint32_t), &nb), "f3ds_get_debug");
            }
        }
        if (merging_type == ADAPTIVE_LAMBDA) lambda = res_.lambda;
    }
    // all_thresh(ground_truth, start, end, step) (:691-741); truth = one ground-truth label per input point (the PCD `label` field).
    // As in the reference: std::out_of_range for bounds outside [0, 1] (:694-698), start > end swapped (:699-705), and the state is
    // left clustered at the LAST threshold of the sweep (:718-726) -- f3ds_auto_threshold leaves the context at the best one (what
    // main() wants next, :432-436), so one more f3ds_recluster follows here.
    std::map<float, performanceSet> all_thresh(const uint32_t* truth_point_labels, float start_thresh, float end_thresh, float step_thresh) {
        need_state("all_thresh");
        if (have_user_) throw std::logic_error("all_thresh needs the frame's points (the ground truth is per input point): use the SupervoxelClustering state");
        if (start_thresh < 0 || start_thresh > 1 || end_thresh < 0 || end_thresh > 1 || step_thresh < 0 || step_thresh > 1)
            throw std::out_of_range("start_thresh, end_thresh and/or step_thresh outside of range [0, 1]");
        const size_t cap = 4096;
        std::vector<float> ts(cap); std::vector<performanceSet> ps(cap);
        size_t n = 0; float bt = 0; performanceSet bp;
        const f3ds_params p = params(0.0f);
        point_labels_.resize(super_->n_);
        check(f3ds_auto_threshold(ctx(), &p, truth_point_labels, start_thresh, end_thresh, step_thresh, ts.data(), ps.data(), cap, &n, &bt, &bp,
                                  point_labels_.data(), 0, &res_), "f3ds_auto_threshold");
        if (merging_type == ADAPTIVE_LAMBDA) lambda = res_.lambda;
        std::map<float, performanceSet> all;
        for (size_t i = 0; i < n && i < cap; ++i) all.insert({ts[i], ps[i]});
        if (!all.empty()) cluster(all.rbegin()->first);
        return all;
    }
    static std::pair<float, performanceSet> best_thresh(const std::map<float, performanceSet>& all) {           // :759-774
        float bt = 0; performanceSet bp = performanceSet();
        for (const auto& kv : all) if (kv.second.fscore > bp.fscore) { bp = kv.second; bt = kv.first; }
        return {bt, bp};
    }
    std::pair<float, performanceSet> best_thresh(const uint32_t* truth_point_labels, float start_thresh, float end_thresh, float step_thresh) {
        return best_thresh(all_thresh(truth_point_labels, start_thresh, end_thresh, step_thresh));
    }
    performanceSet eval_performance(const uint32_t* truth_point_labels) const {      // Testing(get_labeled_cloud(), truth).eval_performance(), main():462-463
        need_state("eval_performance");
        performanceSet p;
        check(f3ds_evaluate(ctx(), truth_point_labels, &p), "f3ds_evaluate");
        return p;
    }
    LabeledCloud get_labeled_cloud() const {                                                                    // :640-663
        need_state("get_labeled_cloud");
        size_t k = 0;
        check(f3ds_get_voxel_cloud(ctx(), nullptr, nullptr, nullptr, 0, &k), "f3ds_get_voxel_cloud");
        LabeledCloud c; c.xyz.resize(3 * k); c.label.resize(k);
        if (k) check(f3ds_get_voxel_cloud(ctx(), c.xyz.data(), c.label.data(), nullptr, k, &k), "f3ds_get_voxel_cloud");
        return c;
    }
    ColoredCloud get_colored_cloud() const {                                                                    // :631-633
        need_state("get_colored_cloud");
        size_t k = 0;
        check(f3ds_get_voxel_cloud(ctx(), nullptr, nullptr, nullptr, 0, &k), "f3ds_get_voxel_cloud");
        ColoredCloud c; c.xyz.resize(3 * k); c.rgba.resize(k);
        if (k) check(f3ds_get_voxel_cloud(ctx(), c.xyz.data(), nullptr, c.rgba.data(), k, &k), "f3ds_get_voxel_cloud");
        return c;
    }
    // get_currentstate().second (:619): adjacency of the merged regions, pairs a < b of surviving supervoxel labels
    std::multimap<uint32_t, uint32_t> get_current_adjacency() const {
        need_state("get_currentstate");
        size_t e = 0;
        check(f3ds_get_region_adjacency(ctx(), nullptr, 0, &e), "f3ds_get_region_adjacency");
        std::vector<uint32_t> pairs(2 * e);
        if (e) check(f3ds_get_region_adjacency(ctx(), pairs.data(), e, &e), "f3ds_get_region_adjacency");
        std::multimap<uint32_t, uint32_t> m;
        for (size_t i = 0; i < e; ++i) m.insert({pairs[2 * i], pairs[2 * i + 1]});
        return m;
    }
    // get_currentstate() (:619-624): state.segments -- the merged regions under the label of their surviving supervoxel -- and weight2adj(state.weight_map)
    std::pair<ClusteringT, AdjacencyMapT> get_currentstate() const {
        need_state("get_currentstate");
        size_t k = 0, nv = 0;
        check(f3ds_get_regions(ctx(), nullptr, nullptr, nullptr, nullptr, nullptr, 0, &k), "f3ds_get_regions");
        std::vector<uint32_t> label(k), cnt(k); std::vector<float> cen(3 * k), nrm(3 * k), rgb(3 * k);
        if (k) check(f3ds_get_regions(ctx(), label.data(), cnt.data(), cen.data(), nrm.data(), rgb.data(), k, &k), "f3ds_get_regions");
        check(f3ds_get_region_voxels(ctx(), nullptr, nullptr, nullptr, 0, &nv), "f3ds_get_region_voxels");
        std::vector<float> xyz(3 * nv); std::vector<uint32_t> rgba(nv), idx(nv);
        if (nv) check(f3ds_get_region_voxels(ctx(), xyz.data(), rgba.data(), idx.data(), nv, &nv), "f3ds_get_region_voxels");
        std::pair<ClusteringT, AdjacencyMapT> ret;
        size_t o = 0;
        for (size_t i = 0; i < k; ++i) {
            Supervoxel& s = ret.first[label[i]];
            for (int a = 0; a < 3; ++a) { s.centroid[a] = cen[3 * i + a]; s.normal[a] = nrm[3 * i + a]; s.mean_rgb[a] = rgb[3 * i + a]; }
            for (uint32_t j = 0; j < cnt[i]; ++j, ++o) { s.voxels.push_back(PointXYZRGBA{xyz[3 * o], xyz[3 * o + 1], xyz[3 * o + 2], rgba[o]}); s.voxel_index.push_back(idx[o]); }
        }
        ret.second = get_current_adjacency();
        return ret;
    }
    // label2color / color2label (clustering.h:207-210, clustering.cpp:793-846): a labelled cloud coloured through the lookup table the coloured cloud uses
    // (f3ds_label_color; opaque alpha), and back -- a label per distinct colour, numbered in order of first appearance.  (The reference keys its map on the
    // FLOAT view of the packed colour, which is a NaN for alpha 255 and red >= 128 and then never compares equal; here the key is the 32 colour bits.)
    static ColoredCloud label2color(const LabeledCloud& label_cloud) {
        ColoredCloud c; c.xyz = label_cloud.xyz; c.rgba.resize(label_cloud.label.size());
        for (size_t i = 0; i < label_cloud.label.size(); ++i) c.rgba[i] = 0xFF000000u | f3ds_label_color(label_cloud.label[i]);
        return c;
    }
    static LabeledCloud color2label(const ColoredCloud& colored_cloud) {
        LabeledCloud c; c.xyz = colored_cloud.xyz; c.label.resize(colored_cloud.rgba.size());
        std::map<uint32_t, uint32_t> mappings;
        uint32_t next = 0;
        for (size_t i = 0; i < colored_cloud.rgba.size(); ++i) {
            const auto it = mappings.find(colored_cloud.rgba[i]);
            if (it != mappings.end()) c.label[i] = it->second;
            else { c.label[i] = next; mappings.insert({colored_cloud.rgba[i], next}); next++; }
        }
        return c;
    }
    // label of the region every input supervoxel ended in (rows in ascending label = the iteration order of segm); set_initialstate(segm, adj) only
    const std::vector<uint32_t>& get_region_of_supervoxel() const { return user_.region_of_sv; }
    // per INPUT POINT region id (composition with pcl getLabeledCloud; F3DS_NO_LABEL for points outside every region); after
    // set_initialstate(segm, adj): per input VOXEL, in the concatenation order of segm's voxels_
    const std::vector<uint32_t>& get_point_labels() const { return point_labels_; }
    const f3ds_result& result() const { return res_; }
};

}  // namespace f3ds
#endif  // F3DS_CLUSTERING_HPP_
