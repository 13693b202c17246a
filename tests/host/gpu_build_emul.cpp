// CPU emulation of the DEVICE structure build (tetra-nerf_amd/csrc/tn_build.hip) -- test infrastructure.
// Every kernel of the device build is a thin loop over an element function of tn_build_core.h; this harness calls the
// same functions from plain loops -- the hash insertions in a SHUFFLED order, because on the GPU the schedule decides
// which sighting of a face claims the table slot -- and requires the products to be byte-identical to the host build
// of tn_mesh.cpp: face table (first-seen order), face -> tets, the 4T walk records, the hull tree.  For the face BVH it
// replays the level-synchronous median-split build (stable sort of (segment, coordinate) keys per level, boxes bottom
// up, greedy 64-wide collapse) and checks the tree's invariants.  Prints "OK ...".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>

#include "tn_build.h"

namespace tn { void set_error(const std::string &) {} }

#define CHECK(c)                                                                      \
    do {                                                                              \
        if (!(c)) { std::fprintf(stderr, "FAILED %s (line %d)\n", #c, __LINE__); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const uint32_t leaf_w = argc > 2 ? (uint32_t)std::atoi(argv[2]) : (uint32_t)tn::WIDE;   // faces per BVH leaf: 16, 32 or 64
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint64_t V = 0, T = 0;
    if (std::fread(&V, 8, 1, f) != 1 || std::fread(&T, 8, 1, f) != 1) return 2;
    std::vector<float> xyz(3 * V);
    std::vector<uint32_t> cells(4 * T);
    if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || std::fread(cells.data(), 4, cells.size(), f) != cells.size()) return 2;
    std::fclose(f);
    using namespace tn;

    // ---- host build (the reference for this test)
    HostMesh hm;
    try { build_face_table(T, cells.data(), hm); } catch (const Error &e) { std::printf("HOST ERROR %s\n", e.what()); }
    // ---- device build, emulated
    const size_t n4 = 4 * T;
    size_t cap = 16;
    while (cap < 8 * T + 16) cap <<= 1;
    std::vector<uint32_t> slot(cap, TN_EMPTY), partner(n4, TN_EMPTY), first(n4), fidx(n4), tet_face(n4, TN_EMPTY);
    uint32_t flags = 0;
    std::vector<uint32_t> order4(n4);
    std::iota(order4.begin(), order4.end(), 0u);
    std::mt19937 rng(12345);
    std::shuffle(order4.begin(), order4.end(), rng);
    for (uint32_t i : order4) core::face_hash_insert(i, cells.data(), slot.data(), cap - 1, partner.data(), &flags);
    if (flags & core::FLAG_TRIPLE_FACE) { std::printf("ERROR A triangle is shared by more than two tetrahedra!\n"); return 3; }
    for (size_t i = 0; i < n4; ++i) first[i] = core::face_is_first((uint32_t)i, partner.data()) ? 1u : 0u;
    std::exclusive_scan(first.begin(), first.end(), fidx.begin(), 0u);
    const size_t F = n4 ? fidx[n4 - 1] + first[n4 - 1] : 0;
    CHECK(F == hm.face_tets.size() / 2);
    std::vector<uint32_t> faces(3 * F), face_tets(2 * F);
    for (uint32_t i : order4)
        if (first[i]) core::face_emit(i, fidx[i], cells.data(), partner.data(), faces.data(), face_tets.data(), tet_face.data());
    CHECK(faces == hm.faces);
    CHECK(face_tets == hm.face_tets);

    // Morton order of the tets + walk records
    std::vector<TetRec> recs;
    std::vector<uint32_t> rec_of_tet_host;
    build_tet_records(T, cells.data(), xyz.data(), hm, recs, rec_of_tet_host);
    std::vector<WalkVar> vars_host;
    build_walk_variants(recs, vars_host);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < T; ++i) {
        float c[3];
        core::tet_centroid((uint32_t)i, cells.data(), xyz.data(), c);
        for (int a = 0; a < 3; ++a) {   // through the order-preserving encoding, as the device reduction does
            lo[a] = core::ordered_float(std::min(core::float_ordered(lo[a]), core::float_ordered(c[a])));
            hi[a] = core::ordered_float(std::max(core::float_ordered(hi[a]), core::float_ordered(c[a])));
        }
    }
    std::vector<std::pair<uint64_t, uint32_t>> keyed(T);
    for (size_t i = 0; i < T; ++i) {
        float c[3];
        core::tet_centroid((uint32_t)i, cells.data(), xyz.data(), c);
        keyed[i] = {core::morton63(c, lo, hi), (uint32_t)i};
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const auto &a, const auto &b) { return a.first < b.first; });   // = the radix sort
    std::vector<uint32_t> order(T), rec_of_tet(T);
    for (size_t r = 0; r < T; ++r) { order[r] = keyed[r].second; rec_of_tet[keyed[r].second] = (uint32_t)r; }
    CHECK(rec_of_tet == rec_of_tet_host);
    // (k_walk_vars, then k_tet_thin / k_thin_patch: star minima in a shuffled order -- the atomic minimum commutes)
    std::vector<WalkVar> vars_dev(n4);
    for (size_t i = 0; i < n4; ++i)
        vars_dev[i] = core::walk_var_of((uint32_t)(i >> 2), (uint32_t)(i & 3), order.data(), rec_of_tet.data(), cells.data(), xyz.data(),
                                        tet_face.data(), faces.data(), face_tets.data(), &flags);
    {
        size_t V = 0;
        for (uint32_t c : cells) V = std::max<size_t>(V, c + 1);
        std::vector<uint32_t> vmin(V, 0x7F800000u);
        for (size_t k = T; k-- > 0;) {
            const uint32_t *c = cells.data() + 4 * k;
            float p[4][3];
            for (int q = 0; q < 4; ++q) for (int a = 0; a < 3; ++a) p[q][a] = xyz[3 * (size_t)c[q] + a];
            const uint32_t bits = core::tet_min_height_bits(p);
            for (int q = 0; q < 4; ++q) core::atomic_min_u32(&vmin[c[q]], bits);
        }
        for (size_t k = 0; k < T; ++k) {
            const uint32_t *c = cells.data() + 4 * k;
            const uint32_t e = core::thin_exponent(vmin[c[0]], vmin[c[1]], vmin[c[2]], vmin[c[3]]);
            for (uint32_t q = 0; q < 4; ++q) vars_dev[4 * (size_t)rec_of_tet[k] + q].code_hi |= e << core::THIN_SHIFT;
        }
    }
    for (size_t i = 0; i < n4; ++i) CHECK(std::memcmp(&vars_dev[i], &vars_host[i], sizeof(WalkVar)) == 0);
    CHECK(!(flags & core::FLAG_INTERNAL));

    // hull tree
    std::vector<uint32_t> hull_ids;
    for (size_t fi = 0; fi < F; ++fi) if (face_tets[2 * fi + 1] == TN_EMPTY) hull_ids.push_back((uint32_t)fi);
    std::vector<float> info(hull_ids.size() * 12);
    for (size_t h = 0; h < hull_ids.size(); ++h)
        core::hull_face_info(hull_ids[h], faces.data(), face_tets.data(), tet_face.data(), rec_of_tet.data(), xyz.data(),
                             reinterpret_cast<uint32_t *>(&info[h * 12]), &flags);
    HostHullBvh hull_dev, hull_host;
    build_hull_from_info(info, hull_dev);
    build_hull_threaded(xyz.data(), hm.faces.data(), hm.face_tets.data(), hull_ids, recs, rec_of_tet_host, hull_host);
    CHECK(hull_dev.nodes.size() == hull_host.nodes.size() && hull_dev.tris.size() == hull_host.tris.size());
    CHECK(std::memcmp(hull_dev.nodes.data(), hull_host.nodes.data(), hull_host.nodes.size() * 4) == 0);
    CHECK(std::memcmp(hull_dev.tris.data(), hull_host.tris.data(), hull_host.tris.size() * 4) == 0);

    // face BVH: level-synchronous median splits
    std::vector<core::BinNode> bn;
    std::vector<std::vector<uint32_t>> frontier;
    std::vector<uint32_t> level_start, leaf_nodes;
    build_bin_topology(F, bn, frontier, level_start, leaf_nodes, leaf_w);
    std::vector<float> fb(6 * F), cen(3 * F);
    for (size_t fi = 0; fi < F; ++fi) core::face_box((uint32_t)fi, faces.data(), xyz.data(), &fb[6 * fi], &cen[3 * fi]);
    std::vector<uint32_t> ord(F);
    std::iota(ord.begin(), ord.end(), 0u);
    for (size_t l = 0; l + 1 < frontier.size(); ++l) {
        const size_t nseg = frontier[l].size();
        std::vector<uint64_t> keys(F);
        for (size_t s = 0; s < nseg; ++s) {
            const core::BinNode &nd = bn[frontier[l][s]];
            float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = nd.first; i < nd.first + nd.count; ++i)
                for (int a = 0; a < 3; ++a) { clo[a] = std::min(clo[a], cen[3 * (size_t)ord[i] + a]); chi[a] = std::max(chi[a], cen[3 * (size_t)ord[i] + a]); }
            const int ax = core::split_axis(clo, chi);
            for (uint32_t i = nd.first; i < nd.first + nd.count; ++i)
                keys[i] = ((uint64_t)s << 32) | core::float_ordered(cen[3 * (size_t)ord[i] + ax]);
        }
        std::vector<uint32_t> perm(F);
        std::iota(perm.begin(), perm.end(), 0u);
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
        std::vector<uint32_t> nxt(F);
        for (size_t i = 0; i < F; ++i) nxt[i] = ord[perm[i]];
        ord.swap(nxt);
    }
    const size_t nn = bn.size();
    std::vector<float> node_lo(3 * nn), node_hi(3 * nn);
    for (size_t k = nn; k-- > 0;) {
        float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (bn[k].left < 0) {
            for (uint32_t i = bn[k].first; i < bn[k].first + bn[k].count; ++i)
                for (int a = 0; a < 3; ++a) { blo[a] = std::min(blo[a], fb[6 * (size_t)ord[i] + a]); bhi[a] = std::max(bhi[a], fb[6 * (size_t)ord[i] + 3 + a]); }
        } else {
            for (int a = 0; a < 3; ++a) {
                blo[a] = std::min(node_lo[3 * (size_t)bn[k].left + a], node_lo[3 * (size_t)bn[k].right + a]);
                bhi[a] = std::max(node_hi[3 * (size_t)bn[k].left + a], node_hi[3 * (size_t)bn[k].right + a]);
            }
        }
        for (int a = 0; a < 3; ++a) { node_lo[3 * k + a] = blo[a]; node_hi[3 * k + a] = bhi[a]; }
    }
    // every face in exactly one leaf; leaves hold <= 64; the topology is a partition
    {
        std::vector<uint8_t> seen(F, 0);
        size_t covered = 0;
        for (uint32_t k : leaf_nodes) {
            CHECK(bn[k].count >= 1 && bn[k].count <= leaf_w && bn[k].left < 0);
            for (uint32_t i = bn[k].first; i < bn[k].first + bn[k].count; ++i) { CHECK(!seen[ord[i]]); seen[ord[i]] = 1; ++covered; }
        }
        CHECK(covered == F);
    }
    // collapse: every binary leaf reachable exactly once, children numbered after their parent
    size_t n_wide = 0, leaf_refs = 0;
    {
        const core::BinTreeView tree{bn.data(), node_lo.data(), node_hi.data()};
        std::vector<int> wide_sub{0};
        for (size_t w = 0; w < wide_sub.size(); ++w) {
            int kids[WIDE];
            const int nk = F ? core::collapse_node(wide_sub[w], tree, kids) : 0;
            CHECK(nk >= (F ? 1 : 0) && nk <= WIDE);
            for (int i = 0; i < nk; ++i) {
                if (bn[kids[i]].left < 0) ++leaf_refs;
                else wide_sub.push_back(kids[i]);
            }
        }
        n_wide = wide_sub.size();
        CHECK(leaf_refs == leaf_nodes.size());
    }
    std::printf("OK faces %zu variants %zu hull %zu bin_nodes %zu leaves %zu wide_nodes %zu\n", F, 4 * (size_t)T, hull_ids.size(), nn,
                leaf_nodes.size(), n_wide);
    return 0;
}
