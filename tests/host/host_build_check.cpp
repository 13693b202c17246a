// Host-side build products of load_tetrahedra checked on the CPU (no GPU, no HIP runtime calls): the face table,
// the adjacency records, the entry-face-specialised walk records, the hull tree and the wide BVH are pure C++
// (tetra-nerf_amd/csrc/tn_mesh.cpp).  Reads a mesh file written by tests/test_host_build.py, checks structural
// invariants, dumps the face table for comparison with the oracle, prints "OK <counts>".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

#include "tn_common.h"

namespace tn { void set_error(const std::string &) {} }

#define CHECK(c)                                                                      \
    do {                                                                              \
        if (!(c)) { std::fprintf(stderr, "FAILED %s (line %d)\n", #c, __LINE__); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint64_t V = 0, T = 0;
    if (std::fread(&V, 8, 1, f) != 1 || std::fread(&T, 8, 1, f) != 1) return 2;
    std::vector<float> xyz(3 * V);
    std::vector<uint32_t> cells(4 * T);
    if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || std::fread(cells.data(), 4, cells.size(), f) != cells.size()) return 2;
    std::fclose(f);

    tn::HostMesh hm;
    tn::build_face_table(T, cells.data(), hm);
    const size_t F = hm.face_tets.size() / 2;
    {
        FILE *o = std::fopen(argv[2], "wb");
        const uint64_t F64 = F;
        std::fwrite(&F64, 8, 1, o);
        std::fwrite(hm.faces.data(), 4, hm.faces.size(), o);
        std::fwrite(hm.face_tets.data(), 4, hm.face_tets.size(), o);
        std::fclose(o);
    }
    std::vector<tn::TetRec> recs;
    std::vector<uint32_t> rec_of_tet;
    tn::build_tet_records(T, cells.data(), xyz.data(), hm, recs, rec_of_tet);
    CHECK(recs.size() == T && rec_of_tet.size() == T);
    // adjacency records: Morton permutation is a bijection, neighbours and back-indices are symmetric
    {
        std::vector<uint8_t> seen(T, 0);
        for (size_t i = 0; i < T; ++i) { CHECK(rec_of_tet[i] < T && !seen[rec_of_tet[i]]); seen[rec_of_tet[i]] = 1; CHECK(recs[rec_of_tet[i]].orig == i); }
        for (size_t r = 0; r < T; ++r)
            for (int k = 0; k < 4; ++k) {
                const tn::TetRec &t = recs[r];
                CHECK(t.vert[k] == cells[4 * (size_t)t.orig + k]);
                const uint32_t fid = t.face[k];
                CHECK(fid < F);
                const uint32_t a = hm.face_tets[2 * (size_t)fid], b = hm.face_tets[2 * (size_t)fid + 1];
                CHECK(a == t.orig || b == t.orig);
                if (t.nbr[k] == TN_EMPTY) { CHECK(b == TN_EMPTY); continue; }
                const uint32_t bk = (t.back >> (2 * k)) & 3u;
                CHECK(recs[t.nbr[k]].nbr[bk] == r && recs[t.nbr[k]].face[bk] == fid);
            }
    }
    // walk records
    std::vector<tn::WalkVar> vars;
    tn::build_walk_variants(recs, vars);
    CHECK(vars.size() == 4 * T);
    size_t hull_exits = 0;
    for (size_t r = 0; r < T; ++r)
        for (uint32_t e = 0; e < 4; ++e) {
            const tn::TetRec &t = recs[r];
            const tn::WalkVar &v = vars[4 * r + e];
            std::set<uint32_t> distinct(t.vert, t.vert + 4);
            if (distinct.size() != 4) continue;  // degenerate tet: codes are arbitrary, the walk flags it
            CHECK(v.orig == t.orig && v.vid[0] == t.vert[e]);
            const uint32_t fe = t.face[e];
            for (int m = 0; m < 3; ++m) { CHECK(v.vid[m + 1] == hm.faces[3 * (size_t)fe + m]); CHECK(v.vid[m + 1] != v.vid[0]); }
            for (int a = 0; a < 3; ++a) CHECK(v.pn[a] == xyz[3 * (size_t)v.vid[0] + a]);
            const uint64_t codes = (uint64_t)v.code_lo | ((uint64_t)v.code_hi << 32);
            for (uint32_t x = 0; x < 3; ++x) {
                const uint32_t fx = v.fid(x);
                CHECK(fx < F && fx != fe);
                // the exit face contains n and the two entry vertices other than a/b/c[x], in ITS stored order = p-code
                const uint32_t code = (uint32_t)(codes >> (12 * x)) & 0xFFFu;
                uint32_t p[3];
                for (int m = 0; m < 3; ++m) {
                    p[m] = (code >> (2 * m)) & 3u;
                    CHECK(p[m] != x + 1);
                    CHECK(v.vid[p[m]] == hm.faces[3 * (size_t)fx + m]);
                }
                for (uint32_t j = 0; j < 3; ++j) {
                    const uint32_t cj = (code >> (6 + 2 * j)) & 3u;
                    if (j == x) CHECK(cj == 3);
                    else CHECK(cj < 3 && p[cj] == j + 1);
                }
                if (v.nb[x] == TN_EMPTY) { CHECK(hm.face_tets[2 * (size_t)fx + 1] == TN_EMPTY); ++hull_exits; continue; }
                CHECK(v.nb[x] < 4 * T);
                const tn::WalkVar &w = vars[v.nb[x]];
                // entering the neighbour through fx: its (a,b,c) is fx's stored triple, its n is not in fx
                for (int m = 0; m < 3; ++m) CHECK(w.vid[m + 1] == hm.faces[3 * (size_t)fx + m]);
                CHECK(recs[v.nb[x] >> 2].face[v.nb[x] & 3u] == fx && (v.nb[x] >> 2) != r);
            }
        }
    // hull tree: every hull face once, with the record / local face that owns it and its vertices in stored order
    std::vector<uint32_t> hull_ids, all(F);
    for (size_t i = 0; i < F; ++i) { all[i] = (uint32_t)i; if (hm.face_tets[2 * i + 1] == TN_EMPTY) hull_ids.push_back((uint32_t)i); }
    tn::HostHullBvh hth;
    tn::build_hull_threaded(xyz.data(), hm.faces.data(), hm.face_tets.data(), hull_ids, recs, rec_of_tet, hth);
    CHECK(hth.tris.size() == hull_ids.size() * 12);
    {
        std::set<uint32_t> seen;
        for (size_t s = 0; s < hull_ids.size(); ++s) {
            uint32_t fid, rec, loc;
            std::memcpy(&fid, &hth.tris[s * 12 + 3], 4); std::memcpy(&rec, &hth.tris[s * 12 + 7], 4); std::memcpy(&loc, &hth.tris[s * 12 + 11], 4);
            CHECK(seen.insert(fid).second && rec < T && loc < 4 && recs[rec].face[loc] == fid && recs[rec].nbr[loc] == TN_EMPTY);
            for (int v = 0; v < 3; ++v)
                for (int a = 0; a < 3; ++a) CHECK(hth.tris[s * 12 + v * 4 + a] == xyz[3 * (size_t)hm.faces[3 * (size_t)fid + v] + a]);
        }
        CHECK(seen.size() == hull_ids.size());
        // threaded tree: skip links point forward, every triangle slot is covered by exactly one leaf
        const size_t nn = hth.nodes.size() / 8;
        std::vector<uint8_t> cov(hull_ids.size(), 0);
        for (size_t i = 0; i < nn; ++i) {
            uint32_t skip, leaf;
            std::memcpy(&skip, &hth.nodes[i * 8 + 3], 4); std::memcpy(&leaf, &hth.nodes[i * 8 + 7], 4);
            CHECK(skip > i && skip <= nn);
            if (leaf != 0xFFFFFFFFu) {
                const uint32_t first = leaf >> 3, cnt = leaf & 7u;
                CHECK(cnt >= 1 && cnt <= 4 && first + cnt <= hull_ids.size());
                for (uint32_t k = 0; k < cnt; ++k) { CHECK(!cov[first + k]); cov[first + k] = 1; }
                for (uint32_t k = 0; k < cnt; ++k)
                    for (int v = 0; v < 3; ++v)
                        for (int a = 0; a < 3; ++a) {
                            const float x = hth.tris[(first + k) * 12 + v * 4 + a];
                            CHECK(x >= hth.nodes[i * 8 + a] && x <= hth.nodes[i * 8 + 4 + a]);
                        }
            }
        }
        for (uint8_t c : cov) CHECK(c);
    }
    // wide BVH: every face in exactly one leaf slot; every internal child box contains its subtree's triangles
    tn::HostWideBvh hb;
    tn::build_wide_bvh(xyz.data(), hm.faces.data(), all, hb);
    {
        const size_t nl = hb.leaf_id.size() / 64;
        std::vector<uint8_t> seen(F, 0);
        for (size_t i = 0; i < hb.leaf_id.size(); ++i)
            if (hb.leaf_id[i] != TN_EMPTY) { CHECK(hb.leaf_id[i] < F && !seen[hb.leaf_id[i]]); seen[hb.leaf_id[i]] = 1; }
        for (uint8_t c : seen) CHECK(c);
        const size_t nn = hb.child.size() / 64;
        CHECK(hb.boxes.size() == nn * 6 * 64 && nn >= 1);
        std::vector<uint8_t> leaf_ref(nl, 0), node_ref(nn, 0);
        node_ref[0] = 1;
        for (size_t n = 0; n < nn; ++n)
            for (int c = 0; c < 64; ++c) {
                const uint32_t ch = hb.child[n * 64 + c];
                if (ch == TN_EMPTY) continue;
                const float *b = &hb.boxes[n * 6 * 64];
                if (ch & 0x80000000u) {
                    const uint32_t li = ch & 0x7FFFFFFFu;
                    CHECK(li < nl && !leaf_ref[li]); leaf_ref[li] = 1;
                    for (int s = 0; s < 64; ++s) {
                        if (hb.leaf_id[li * 64 + s] == TN_EMPTY) continue;
                        for (int k = 0; k < 9; ++k) {
                            const float x = hb.leaf_tri[(li * 9 + k) * 64 + s];
                            CHECK(x >= b[(k % 3) * 64 + c] && x <= b[(3 + k % 3) * 64 + c]);
                        }
                    }
                } else {
                    CHECK(ch < nn && ch > n && !node_ref[ch]); node_ref[ch] = 1;
                }
            }
        for (uint8_t c : leaf_ref) CHECK(c);
        for (uint8_t c : node_ref) CHECK(c);
    }
    std::printf("OK faces %zu hull %zu variants %zu hull_exits %zu\n", F, hull_ids.size(), vars.size(), hull_exits);
    return 0;
}
