// pcl_dump.cpp -- for a maintainer WITH PCL >= 1.8 and OpenCV: dump what the reference really computes, in the array format of
// f3ds_get_debug / f3ds_oracle_get, so that tests/pcl_pin/compare_with_pcl.py can pin the half of the path this repo could only restate
// (SURVEY.md 8c "parity unpinned": voxel order, centroids, normals, supervoxel labels, adjacency, OpenCV's Lab, the Glasbey table).
//
// NOT BUILT OR RUN IN THIS REPO'S IMAGE (no PCL, no OpenCV, no network): written against the PCL 1.10 / OpenCV 4 headers from memory, never
// compiled here.  It is test infrastructure for a machine that has the reference's dependencies; nothing in the product includes it.
//
// Build (inside a checkout of the reference, next to its own sources -- it links the reference's clustering.cpp / color_utilities.cpp):
//   g++ -O2 -std=c++14 -I<reference>/include pcl_dump.cpp <reference>/src/clustering.cpp <reference>/src/clustering_state.cpp \
//       <reference>/src/color_utilities.cpp $(pkg-config --cflags --libs pcl_segmentation-1.10 pcl_io-1.10 pcl_features-1.10 opencv4) -o pcl_dump
// Run:   ./pcl_dump <frame.pcd> <out dir> [-v 0.008] [-s 0.08] [-c 0.2] [-z 0.4] [-n 1.0] [--NT] [--RGB] [--CVX] [--ML x | --AL | --EQ b] [-t 0.2]
//        (the flags of the reference's CLI, src/supervoxel_clustering.cpp:187-298; defaults as there)
// Then:  python tests/pcl_pin/compare_with_pcl.py <out dir> <frame.pcd>    (in THIS repo)
//
// What is written (<NAME>.bin raw little-endian + manifest.txt "NAME dtype count"); leaf order = adjacency_octree_->begin()..end(), the
// order VoxelData::idx_ numbers (the order of F3DS_DBG_VOXEL_*):
//   GRID f64[5]             bounding box min x,y,z, resolution, tree depth (OctreePointCloud::getBoundingBox / getResolution / getTreeDepth)
//   VOXEL_COUNT u32[V]      points per leaf (LeafContainerT::getPointCounter)
//   VOXEL_XYZ f32[V*3], VOXEL_RGB f32[V*3], VOXEL_NORMAL f32[V*4], VOXEL_DIST f32[V]    VoxelData::xyz_, rgb_, normal_, distance_
//   VOXEL_NEIGHBOR_LIST i32[V*27]   leaf ordinals in the order of the leaf's neighbour list, -1 padded (compared as a set per voxel with VOXEL_NEIGHBORS)
//   VOXEL_SVLABEL u32[V]    supervoxel label per leaf, recovered from getLabeledVoxelCloud() by exact xyz match (SupervoxelHelper is private)
//   POINT_SVLABEL u32[N]    getLabeledCloud(): supervoxel label per input point (0 = none)
//   SV_LABELS u32[S], SV_CENTROID f32[S*10]    the supervoxel_clusters map: key; centroid_ xyz, centroid_ r g b, normal_ x y z, 0
//   SV_VOXEL_OFFSET u32[S+1], SV_VOXEL_XYZ f32[Vt*3], SV_VOXEL_RGBA u32[Vt]      voxels_ of every supervoxel (the f3ds_supervoxel_set arrays)
//   ADJACENCY u32[P*2]      getSupervoxelAdjacency() multimap in iteration order (both directions): the input of f3ds_cluster_supervoxels
//   LAMBDA f32[1]           Clustering::get_lambda() after cluster()
//   CLOUD_XYZ f32[K*3], CLOUD_LABEL u32[K]     Clustering::get_labeled_cloud(): the order and ids of f3ds_get_voxel_cloud
//   REGION_LABELS u32[R], REGION_COUNT u32[R], REGION_CENTROID f32[R*3], REGION_NORMAL f32[R*3]   get_currentstate().first
//   REGION_ADJACENCY u32[A*2]                  get_currentstate().second
//   LAB17 f32[17*17*17*3]   ColorUtilities::rgb2lab on the lattice r,g,b in {0,16,...,240,255} (OpenCV's float Lab: pins SURVEY a13)
//   GLASBEY u32[256]        ColorUtilities::get_glasbey(i) as 0x00RRGGBB (pcl::GlasbeyLUT: pins the colours of get_colored_cloud)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include <pcl/io/pcd_io.h>
#include <pcl/point_types.h>
#include <pcl/segmentation/supervoxel_clustering.h>

#include "supervoxel_clustering/clustering.h"
#include "supervoxel_clustering/color_utilities.h"

typedef pcl::PointXYZRGBA PointT;
typedef pcl::PointXYZRGBL PointLCT;

// the octree and its leaves are protected members of pcl::SupervoxelClustering: reached through a derived class
struct Probe : public pcl::SupervoxelClustering<PointT> {
    typedef pcl::SupervoxelClustering<PointT> Base;
    Probe(float v, float s) : Base(v, s) {}
    typename Base::OctreeAdjacencyT& octree() { return *this->adjacency_octree_; }
};

static std::string g_dir;
static std::ofstream g_manifest;
template <class T> static void put(const char* name, const char* dtype, const std::vector<T>& v) {
    std::ofstream f(g_dir + "/" + name + ".bin", std::ios::binary);
    if (!v.empty()) f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
    g_manifest << name << " " << dtype << " " << v.size() << "\n";
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: pcl_dump <frame.pcd> <out dir> [reference CLI flags]\n"); return 1; }
    const std::string pcd = argv[1];
    g_dir = argv[2];
    float voxel_res = 0.008f, seed_res = 0.08f, w_c = 0.2f, w_s = 0.4f, w_n = 1.0f, lambda = 0, thresh = 0.2f;
    bool nt = false, rgb = false, cvx = false, ml = false, eq = false; int bins = 0;
    for (int i = 3; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() { return i + 1 < argc ? std::atof(argv[++i]) : 0.0; };
        if (a == "-v") voxel_res = (float)val(); else if (a == "-s") seed_res = (float)val(); else if (a == "-c") w_c = (float)val();
        else if (a == "-z") w_s = (float)val(); else if (a == "-n") w_n = (float)val(); else if (a == "-t") thresh = (float)val();
        else if (a == "--NT") nt = true; else if (a == "--RGB") rgb = true; else if (a == "--CVX") cvx = true; else if (a == "--AL") {}
        else if (a == "--ML") { ml = true; lambda = (float)val(); } else if (a == "--EQ") { eq = true; bins = (int)val(); }
    }
    g_manifest.open(g_dir + "/manifest.txt");

    // ---- main():313-340
    pcl::PointCloud<PointLCT>::Ptr input(new pcl::PointCloud<PointLCT>);
    if (pcl::io::loadPCDFile(pcd, *input)) return 2;
    for (auto& p : input->points) if (p.z < 0) p.z = std::abs(p.z);
    pcl::PointCloud<PointT>::Ptr cloud(new pcl::PointCloud<PointT>);
    pcl::copyPointCloud(*input, *cloud);

    // ---- main():348-367
    Probe super(voxel_res, seed_res);
    super.setUseSingleCameraTransform(!nt);
    super.setInputCloud(cloud);
    super.setColorImportance(w_c); super.setSpatialImportance(w_s); super.setNormalImportance(w_n);
    std::map<uint32_t, pcl::Supervoxel<PointT>::Ptr> clusters;
    super.extract(clusters);
    pcl::PointCloud<pcl::PointXYZL>::Ptr full_labeled = super.getLabeledCloud();
    pcl::PointCloud<pcl::PointXYZL>::Ptr voxel_labeled = super.getLabeledVoxelCloud();
    std::multimap<uint32_t, uint32_t> adjacency;
    super.getSupervoxelAdjacency(adjacency);

    auto& oct = super.octree();
    {
        double x0, y0, z0, x1, y1, z1;
        oct.getBoundingBox(x0, y0, z0, x1, y1, z1);
        put("GRID", "f64", std::vector<double>{x0, y0, z0, oct.getResolution(), (double)oct.getTreeDepth()});
    }
    std::unordered_map<const void*, int> ordinal;
    { int i = 0; for (auto it = oct.begin(); it != oct.end(); ++it) ordinal[(const void*)*it] = i++; }
    const size_t V = ordinal.size();
    std::vector<uint32_t> vcount, vlabel(V, 0u); std::vector<float> vxyz, vrgb, vnrm, vdist; std::vector<int32_t> vnbr(V * 27, -1);
    struct Key { float x, y, z; bool operator<(const Key& o) const { return std::memcmp(this, &o, sizeof(Key)) < 0; } };
    std::map<Key, uint32_t> label_of_xyz;
    for (const auto& p : voxel_labeled->points) label_of_xyz[Key{p.x, p.y, p.z}] = p.label;
    size_t li = 0;
    for (auto it = oct.begin(); it != oct.end(); ++it, ++li) {
        auto* leaf = *it;
        const auto& d = leaf->getData();
        vcount.push_back((uint32_t)leaf->getPointCounter());
        for (int a = 0; a < 3; ++a) { vxyz.push_back(d.xyz_[a]); vrgb.push_back(d.rgb_[a]); }
        for (int a = 0; a < 4; ++a) vnrm.push_back(d.normal_[a]);
        vdist.push_back(d.distance_);
        int k = 0;
        for (auto nb = leaf->cbegin(); nb != leaf->cend() && k < 27; ++nb, ++k) vnbr[li * 27 + k] = ordinal.at((const void*)*nb);
        auto f = label_of_xyz.find(Key{d.xyz_[0], d.xyz_[1], d.xyz_[2]});
        if (f != label_of_xyz.end()) vlabel[li] = f->second;
    }
    put("VOXEL_COUNT", "u32", vcount); put("VOXEL_XYZ", "f32", vxyz); put("VOXEL_RGB", "f32", vrgb); put("VOXEL_NORMAL", "f32", vnrm);
    put("VOXEL_DIST", "f32", vdist); put("VOXEL_NEIGHBOR_LIST", "i32", vnbr); put("VOXEL_SVLABEL", "u32", vlabel);
    { std::vector<uint32_t> pl; for (const auto& p : full_labeled->points) pl.push_back(p.label); put("POINT_SVLABEL", "u32", pl); }

    std::vector<uint32_t> svl, svoff{0}, svrgba; std::vector<float> svc, svxyz;
    for (const auto& kv : clusters) {
        const auto& s = *kv.second;
        svl.push_back(kv.first);
        svc.insert(svc.end(), {s.centroid_.x, s.centroid_.y, s.centroid_.z, (float)s.centroid_.r, (float)s.centroid_.g, (float)s.centroid_.b,
                               s.normal_.normal_x, s.normal_.normal_y, s.normal_.normal_z, 0.0f});
        for (const auto& p : s.voxels_->points) { svxyz.insert(svxyz.end(), {p.x, p.y, p.z}); svrgba.push_back((uint32_t)p.r << 16 | (uint32_t)p.g << 8 | (uint32_t)p.b); }
        svoff.push_back((uint32_t)svrgba.size());
    }
    put("SV_LABELS", "u32", svl); put("SV_CENTROID", "f32", svc); put("SV_VOXEL_OFFSET", "u32", svoff); put("SV_VOXEL_XYZ", "f32", svxyz); put("SV_VOXEL_RGBA", "u32", svrgba);
    { std::vector<uint32_t> adj; for (const auto& kv : adjacency) { adj.push_back(kv.first); adj.push_back(kv.second); } put("ADJACENCY", "u32", adj); }

    // ---- main():408-449
    Clustering seg;
    if (rgb) seg.set_delta_c(RGB_EUCL);
    if (cvx) seg.set_delta_g(CONVEX_NORMALS_DIFF);
    if (ml) { seg.set_merging(MANUAL_LAMBDA); if (lambda != 0) seg.set_lambda(lambda); }
    else if (eq) { seg.set_merging(EQUALIZATION); if (bins != 0) seg.set_bins_num((short)bins); }
    seg.set_initialstate(clusters, adjacency);
    seg.cluster(thresh);
    put("LAMBDA", "f32", std::vector<float>{seg.get_lambda()});
    {
        auto lc = seg.get_labeled_cloud();
        std::vector<float> xyz; std::vector<uint32_t> lab;
        for (const auto& p : lc->points) { xyz.insert(xyz.end(), {p.x, p.y, p.z}); lab.push_back(p.label); }
        put("CLOUD_XYZ", "f32", xyz); put("CLOUD_LABEL", "u32", lab);
    }
    {
        auto st = seg.get_currentstate();
        std::vector<uint32_t> rl, rc, ra; std::vector<float> rcen, rn;
        for (const auto& kv : st.first) {
            rl.push_back(kv.first); rc.push_back((uint32_t)kv.second->voxels_->size());
            rcen.insert(rcen.end(), {kv.second->centroid_.x, kv.second->centroid_.y, kv.second->centroid_.z});
            rn.insert(rn.end(), {kv.second->normal_.normal_x, kv.second->normal_.normal_y, kv.second->normal_.normal_z});
        }
        for (const auto& kv : st.second) { ra.push_back(kv.first); ra.push_back(kv.second); }
        put("REGION_LABELS", "u32", rl); put("REGION_COUNT", "u32", rc); put("REGION_CENTROID", "f32", rcen); put("REGION_NORMAL", "f32", rn); put("REGION_ADJACENCY", "u32", ra);
    }
    {
        std::vector<float> lab;
        const int lv[17] = {0, 16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 255};
        for (int r : lv) for (int g : lv) for (int b : lv) {
            float c[3] = {(float)r, (float)g, (float)b};
            float* o = ColorUtilities::rgb2lab(c);
            lab.insert(lab.end(), {o[0], o[1], o[2]});
            delete[] o;
        }
        put("LAB17", "f32", lab);
        std::vector<uint32_t> gl;
        for (uint32_t i = 0; i < 256; ++i) { uint8_t* c = ColorUtilities::get_glasbey(i); gl.push_back((uint32_t)c[0] << 16 | (uint32_t)c[1] << 8 | c[2]); delete[] c; }
        put("GLASBEY", "u32", gl);
    }
    std::printf("pcl_dump: V %zu, S %zu, regions written to %s\n", V, clusters.size(), g_dir.c_str());
    return 0;
}
