// tn_api.hip -- the C-ABI of libtetranerf_hip.so (see include/tetranerf_hip.h).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

#include "../../include/tetranerf_hip.h"
#include "tn_build.h"
#include "tn_common.h"
#include "tn_devbuf.h"
#include "tn_kernels.h"

namespace tn {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

}  // namespace tn

struct tn_tracer {
    int device = 0;
    tn::DeviceMesh mesh;
    tn::HostMesh host;  // kept for tn_get_faces (device build: downloaded on first use)
    bool gpu_build = true;   // structures built on the device (tn_build.hip); false: the single-threaded host build (tn_mesh.cpp)
    uint32_t bvh_max_stack = 1;
    unsigned leaf_width = 16;            // faces per BVH leaf block (16 / 32 / 64; applies at the next load_tetrahedra)
    tn::DevBuf<uint32_t> faces, face_tets, fallback_list, walk_n;
    tn::DevBuf<uint4> hull_entry;        // [R] k_hull_entry -> k_trace_walk
    tn::DevBuf<uint2> literal_list;      // rays whose logged hits go through the literal sort + pairing
    tn::DevBuf<uint4> hit_log;           // walk -> segment writer / literal pairing: 16 B per recorded hit, [rays / 64][M][64]
    size_t log_cap_bytes = 0;            // 0: a fraction of the free device memory (decided per call); larger calls are walked
                                         // + written in ray chunks.  Option log_cap_mb (tests)
    bool literal = true;                 // false: rays with uncertified order are re-traced through the BVH instead of being
                                         // paired from the log (cross-check of the two paths; tests)
    hipStream_t side = nullptr;          // literal pairing of the logged hits (beside the tail fill)
    hipStream_t aux = nullptr;           // BVH re-trace of the fallback rays (forked right after the walk)
    hipStream_t pre = nullptr;           // speculative tail fill beside the walk
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_start = nullptr, ev_pre = nullptr, ev_seg = nullptr, ev_aux = nullptr;
    int spec_fill = 0;                   // 1: the last quarter of every row is filled beside the walk where that is mesh-safe; 0 off (default since round 6)
    unsigned spec_k0 = 0;                // override of the first speculatively filled slot (multiple of 32; tests)
    // Round 4: the speculative fill used to be launched with 2048 blocks = 32 waves per CU -- every wave slot of the chip --
    // in front of a walk that wants all 32 slots itself (64 VGPRs): the kernels shared the time instead of overlapping.  Now
    // the fill holds 2 blocks = 8 waves per CU (enough for the write ceiling) and the walk is limited to 6 blocks = 24 waves
    // per CU by a 26 KB dynamic-LDS reservation while a fill runs beside it, so both are resident for the walk's whole
    // duration: -3.1 % on the C2 frame, -4.9 % on the C4 frame (interleaved sweep on one box, profiles/r04f_overlap_sweep.txt;
    // 7 or 5 walk blocks, 1 or 4 fill blocks per CU are all worse).  Options for sweeps:
    unsigned spec_blocks = 512;          // grid of the speculative fill
    bool hull_flat = true;               // the walk finds its hull faces through the flat box table in LDS (hulls of <= 1024 faces); false: threaded tree
    unsigned writer_blocks = 0;          // grid of the segment writer (0: 2 blocks per CU, what is resident at once)
    // Round 6: the tail fill is cut fine (one block per row, k_fill_rows_fine) and nothing is filled beside the walk any more:
    // -5.9 / -6.6 / -1.3 % on the C2 / C4 frames / C5 rays, averaged over fresh allocations of the rows in one process
    // (profiles/r06u_alloc_sweep.txt; persistent waves at 512 ... 160000 blocks: r06t_alloc_sweep*.txt).  With the round's
    // faster writer the overlap of fill and walk had stopped paying (r06r_spec_sweep.txt: on / off +-0.4 %).
    unsigned fill_blocks = tn::FILL_FINE; // grid of the tail fill (option "fill_blocks": -2 = one linear stream per array, -1 = one block per row, else persistent waves)
    unsigned walk_lds_kb = 26;           // dynamic LDS reserved per walk block beside a speculative fill (0: no limit)
    bool small_lds = true;               // small batches: LDS hit arrays sized for the mesh, overflow rays in a second launch
    unsigned lds_cap = 0;                // 0: from the mesh size; otherwise the entries of the small arrays (power of two; tests)
    bool dense_tails = true;             // false: slots >= num_visited stay unwritten on walked rows (non-reference, compact use)
    // The walk's certification has an unproved residue, so certified rays are cross-checked against a
    // count-only BVH all-hits traversal, on the aux stream beside the writer and the fill (one-chunk calls).  Two populations:
    //   * the RISK classes (round 5): every certified ray inside the wide band of a guard (walk: edge_band; option "risk_band" = 2:
    //     between 8 and 16 rounding distances of a hull edge / of an edge of a thin-neighbourhood tet) -- 0.3-0.5 % of the rays;
    //   * a blind sample: every verify_stride-th certified ray.
    // Measured interleaved in one process (profiles/r05e_risk_sweep.txt; C2 frame / C4 frame / C5 rays): round 4's blind sample
    // of 1 in 256 cost +0.8 / +0.9 / +0.0 % over no check at all; the risk classes at band 2 + a blind 1 in 1024 cost the same
    // (+0.8 / +1.3 / -0.0 %) while checking EVERY ray of the classes (1,850 / 2,705 / 5,176 rays) and 618 / 609 / 924 blind ones;
    // band 4 costs +1.1 / +5.3 / +4.1 %, band 8 +5.5 / +13 / +15 % (three to seven times as many rays).  Hence band 2.
    // Round 6: the blind stride is back at 256 beside the risk classes (the residue of the certification is not proved, and the
    // blind sample is all that looks beyond the band): in-process sweep, 1024 -> 256: +0.0 / +0.7 / +1.1 %, 64: +2.1 / +1.7 / +6.7 %
    // (profiles/r06i_stride_sweep.txt) -- paid for by rules A-C of the order test (-0.1 / -1.9 / -3.5 % in the same sweep).
    unsigned verify_stride = 256;
    unsigned literal_sort_passes = 8;    // odd-even passes over a literal ray's logged hits before the bitonic network (tests: 0, 1)
    bool verify_inject = false;          // tests: every cross-checked ray is treated as a mismatch (exercises the hand-over)
    tn::DevBuf<uint32_t> verify_list;    // certified rays whose count differed: re-traced by the BVH kernel at the end of the call
    tn::DevBuf<tn::WalkVar> vars;        // the build's 64-byte records: split into the three tables below, then released
    tn::DevBuf<tn::WalkHot> hot;
    tn::DevBuf<tn::WalkCold> cold;
    tn::DevBuf<tn::WalkTet> tets;
    int writer_table = 0;                // 0: by mesh size (WALK_TET_MIN_TETS), 1: per (tet, entry face), 2: per tet (tests, A/B)
    tn::DevBuf<tn::WalkFid> fidt;
    tn::DevBuf<float> hull_nodes, hull_tris;
    tn::DevWideBvh bvh;
    static constexpr int N_STATS = 32;      // 64-bit counters of the last call: [0..4) path statistics, [4..20) walk hand-over reasons,
                                            // [20..24) diagnostics, [24..28) risk classes of the certification (tn_trace_cross_check)
    static constexpr int N_CTR = 2;         // 64-bit words behind the statistics: four uint32 device-side counters
    tn::DevBuf<unsigned long long> stats;   // [N_STATS] counters + [N_STATS .. N_STATS + N_CTR) uint32: fallback count,
                                            // literal count, kmax (one memset clears them all)
    uint32_t *fallback_count() { return reinterpret_cast<uint32_t *>(stats.p + N_STATS); }
    uint32_t *literal_count() { return reinterpret_cast<uint32_t *>(stats.p + N_STATS) + 1; }
    uint32_t *verify_count() { return reinterpret_cast<uint32_t *>(stats.p + N_STATS) + 2; }
    uint32_t *risk_count() { return reinterpret_cast<uint32_t *>(stats.p + N_STATS) + 3; }
    // Round 6, measured and dropped (profiles/r06d_lib_ab.txt, r06e_sweep.txt, r06f_sweep.txt; the code is in the history:
    // commit "wip: pipelined segment writer"): the segment writer as a three-stage software pipeline with unconditional memory
    // instructions over a walk-built list of non-empty groups.  Alone it equals the grouped-store writer below (0.62 ms on the C2
    // frame), in the schedule it costs +3..6 %: it leaves the padding [n, ceil32(n)) to the tail fill, and a line written in part
    // by two kernels costs the fill 12-18 % (partial-line writes).  The tail fill BESIDE that writer: +17 % / +3 % / +0..6 %.
    // The walk's order test (tn_trace_walk.hip): 0 = round 5's pairwise test (OrderR5), 3 = the same + the end-of-chain rules A-C
    // (OrderR5e), 1 = round 6's cluster test (OrderR6: rules A-D), 2 = by mesh size (default).  Same rows whichever is used (the
    // literal kernel writes what the writer does not).  In-process sweeps (profiles/r06f_sweep.txt, r06i_stride_sweep.txt: A-C
    // against round 5: C2 +-0, C4 -1.9..-2.8 %, C5 -3.5..-3.9 %; r06l_sweep.txt, r06m_sweep*.txt, r06p_place_sweep.txt: the cluster
    // test against round 5: C2 +2.5..3.8 %, C4 +1.1..2.6 %, C5 -1.8..-2.9 %, i.e. its extra state costs the frames ~4 % where the
    // walk is VALU-bound beside the speculative fill and pays where 9-27 % of the rays would be literal): hence A-C below
    // WALK_TET_MIN_TETS tets, the cluster test from there on.
    int cert_ends = 2;

    tn::DevBuf<uint32_t> risk_list;      // certified rays inside the wide band of a certification guard (all cross-checked)
    bool verify_risk = true;             // option "verify_risk"
    unsigned risk_band = 2;              // option "risk_band": width of the risk classes' band, in units of the guards' 8 delta
    size_t last_num_rays = 0;
    int use_walk = 1;                    // 0 never, 1 from walk_min_rays rays on, 2 always
    size_t walk_min_rays = 12288;        // measured crossover on the 100k ... 1M-tet meshes (round 6 again: profiles/r06y_batch_crossover.txt)
    bool walk_min_auto = true;           // ... and by mesh size above that (until option "walk_min_rays" is set): the BVH path's LDS hit arrays grow
                                         // with the mesh, a batch then needs several rounds of waves: 8192 from 2M tets, 6144 from 4M tets on
                                         // (2.7 M / 6.7 M tets at 8192 rays: BVH 2.11 / 3.37 ms, walk 2.02 / 2.68; profiles/r06al_big_mesh_batches.txt)
                                         // (profiles/r02t_crossover.txt: 8192 rays 0.46-0.49 vs 0.61-0.71 ms, 12288 rays 0.68-0.90 vs
                                         //  0.67-0.82 ms, 16384 rays 0.89-1.16 vs 0.67-0.84 ms; round 2a: 6144)
    bool last_walk = false;
    bool loaded = false;
    hipStream_t last_stream = nullptr;
    // One tracer = one set of scratch buffers, counters, side streams and events: calls on the SAME handle are serialised
    // here (host section only: the kernels of two calls still queue behind each other on their streams).  ctypes releases
    // the GIL, so a viewer thread and a trainer sharing a tracer can be inside tn_trace_rays* at the same time.
    std::mutex mu;
    // option "timing" = 1: every kernel of a one-chunk walk call is enqueued on the CALLER's stream, in program order,
    // with a timing event after each -- the per-kernel breakdown bench.py prints (tn_trace_timings); the schedule of a
    // normal call overlaps them on four streams, so the parts do not add up to a call's duration
    bool timing = false;
    static constexpr int N_TEV = 9;
    hipEvent_t tev[N_TEV] = {};
    bool tev_valid = false;
};

namespace {

template <typename Fn>
int guarded(Fn &&fn) {
    try {
        fn();
        tn::set_error("");
        return 0;
    } catch (const std::exception &e) {
        tn::set_error(e.what());
        return 1;
    } catch (...) {
        tn::set_error("unknown error");
        return 1;
    }
}

struct DeviceGuard {
    int prev = 0;
    explicit DeviceGuard(int dev) {
        TN_HIP(hipGetDevice(&prev));
        if (prev != dev) TN_HIP(hipSetDevice(dev));
        cur = dev;
    }
    ~DeviceGuard() {
        if (prev != cur) (void)hipSetDevice(prev);
    }
    int cur;
};

tn_tracer *checked(tn_tracer_t t) {
    if (!t) throw tn::Error("tracer handle is null");
    return t;
}

bool env_flag(const char *name, bool dflt) {
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    return !(v[0] == '0' || v[0] == 'n' || v[0] == 'N' || v[0] == 'f' || v[0] == 'F');
}

}  // namespace

extern "C" {

const char *tn_last_error(void) { return tn::g_last_error.c_str(); }

#define TN_STR2(x) #x
#define TN_STR(x) TN_STR2(x)
const char *tn_version(void) { return "tetranerf_hip 0.6.0 abi " TN_STR(TN_ABI_VERSION) " gfx950"; }
int tn_abi_version(void) { return TN_ABI_VERSION; }

int tn_tracer_create(int device, tn_tracer_t *out) {
    return guarded([&] {
        if (!out) throw tn::Error("out is null");
        int count = 0;
        TN_HIP(hipGetDeviceCount(&count));
        if (device < 0 || device >= count) throw tn::Error("The device argument must be a CUDA device.");
        DeviceGuard g(device);
        auto t = std::make_unique<tn_tracer>();
        t->device = device;
        t->use_walk = env_flag("TETRANERF_HIP_WALK", true) ? 1 : 0;
        t->gpu_build = env_flag("TETRANERF_HIP_GPU_BUILD", true);
        t->stats.alloc(tn_tracer::N_STATS + tn_tracer::N_CTR);
        TN_HIP(hipMemset(t->stats.p, 0, (tn_tracer::N_STATS + tn_tracer::N_CTR) * sizeof(unsigned long long)));
        {
            // the side streams carry the few rays the walk does not certify: lowest priority, so that the dispatcher
            // hands wave slots to the main stream's kernels first when both have blocks waiting
            int least = 0, greatest = 0;
            TN_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            TN_HIP(hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, least));
            TN_HIP(hipStreamCreateWithPriority(&t->aux, hipStreamNonBlocking, least));
        }
        TN_HIP(hipStreamCreateWithFlags(&t->pre, hipStreamNonBlocking));
        for (hipEvent_t *e : {&t->ev_fork, &t->ev_join, &t->ev_start, &t->ev_pre, &t->ev_seg, &t->ev_aux})
            TN_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        *out = t.release();
    });
}

int tn_tracer_destroy(tn_tracer_t tracer) {
    return guarded([&] {
        if (!tracer) return;
        DeviceGuard g(tracer->device);
        (void)hipDeviceSynchronize();
        for (hipStream_t st : {tracer->side, tracer->aux, tracer->pre})
            if (st) (void)hipStreamDestroy(st);
        for (hipEvent_t e : {tracer->ev_fork, tracer->ev_join, tracer->ev_start, tracer->ev_pre, tracer->ev_seg, tracer->ev_aux})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : tracer->tev)
            if (e) (void)hipEventDestroy(e);
        delete tracer;
    });
}

int tn_load_tetrahedra(tn_tracer_t tracer, size_t V, size_t T, const float *xyz, const uint32_t *cells,
                       void *stream_) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        std::lock_guard<std::mutex> lock(t->mu);
        DeviceGuard g(t->device);
        hipStream_t stream = (hipStream_t)stream_;
        if ((V && !xyz) || (T && !cells)) throw tn::Error("xyz / cells must not be null");
        if (V >= 0xFFFFFFFFull || T >= 0x0FFFFFFFull) throw tn::Error("mesh too large (uint32 ids)");
        t->loaded = false;
        t->host.faces.clear(); t->host.face_tets.clear();
        size_t F = 0, n_hull = 0, n_hull_nodes = 0;
        if (t->gpu_build && T > 0) {
            // everything is built on the device from the caller's buffers (tn_build.hip)
            tn::BuildInfo bi;
            tn::device_build(V, T, xyz, cells, stream,
                             tn::BuildTargets{t->faces, t->face_tets, t->vars, t->hull_nodes, t->hull_tris, t->bvh}, bi, t->leaf_width);
            // (+ WIDE: the traversal pops the next node before it pushes the current one's children)
            if (bi.max_stack + (uint32_t)tn::WIDE > (uint32_t)tn::STACK_CAP)
                throw tn::Error("face BVH too deep for the traversal stack (" + std::to_string(bi.max_stack) + " > " +
                                std::to_string(tn::STACK_CAP) + " entries)");
            t->host.scene_max = bi.scene_max;
            t->bvh_max_stack = bi.max_stack;
            F = bi.F; n_hull = bi.n_hull; n_hull_nodes = bi.n_hull_nodes;
        } else {
        // host build: blocking D2H of the mesh (the reference does the same: tetrahedra_tracer.cpp:255-259)
        std::vector<float> hxyz(3 * V);
        std::vector<uint32_t> hcells(4 * T);
        TN_HIP(hipStreamSynchronize(stream));
        if (V) TN_HIP(hipMemcpy(hxyz.data(), xyz, hxyz.size() * sizeof(float), hipMemcpyDeviceToHost));
        if (T) TN_HIP(hipMemcpy(hcells.data(), cells, hcells.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < hcells.size(); ++i)
            if (hcells[i] >= V) throw tn::Error("cells contains a vertex index that is out of bounds");

        tn::build_face_table(T, hcells.data(), t->host);
        F = t->host.face_tets.size() / 2;
        float smax = 0.f;
        for (size_t i = 0; i < hcells.size(); ++i)
            for (int k = 0; k < 3; ++k) smax = std::max(smax, std::fabs(hxyz[3 * (size_t)hcells[i] + k]));
        t->host.scene_max = smax;

        std::vector<uint32_t> all(F), hull_ids;
        for (size_t f = 0; f < F; ++f) {
            all[f] = (uint32_t)f;
            if (t->host.face_tets[2 * f + 1] == TN_EMPTY) hull_ids.push_back((uint32_t)f);
        }
        tn::HostWideBvh hb;
        tn::build_wide_bvh(hxyz.data(), t->host.faces.data(), all, hb, t->leaf_width);
        if (hb.max_stack + (uint32_t)tn::WIDE > (uint32_t)tn::STACK_CAP)
            throw tn::Error("face BVH too deep for the traversal stack (" + std::to_string(hb.max_stack) + " > " +
                            std::to_string(tn::STACK_CAP) + " entries)");
        t->bvh_max_stack = hb.max_stack;
        std::vector<tn::TetRec> recs;
        std::vector<uint32_t> rec_of_tet;
        tn::build_tet_records(T, hcells.data(), hxyz.data(), t->host, recs, rec_of_tet);
        tn::HostHullBvh hth;
        tn::build_hull_threaded(hxyz.data(), t->host.faces.data(), t->host.face_tets.data(), hull_ids, recs, rec_of_tet, hth);

        t->faces.upload(t->host.faces);
        t->face_tets.upload(t->host.face_tets);
        t->bvh.upload(hb, smax);
        {
            std::vector<tn::WalkVar> vars;
            tn::build_walk_variants(recs, vars);
            t->vars.upload(vars);
        }
        t->hull_nodes.upload(hth.nodes_and_flat());
        t->hull_tris.upload(hth.tris);
        n_hull = hull_ids.size(); n_hull_nodes = hth.nodes.size() / 8;
        }

        tn::DeviceMesh &m = t->mesh;
        m.xyz = xyz; m.cells = cells;
        m.V = (uint32_t)V; m.T = (uint32_t)T; m.F = (uint32_t)F;
        m.faces = t->faces.p; m.face_tets = t->face_tets.p;
        m.bvh = t->bvh.view;
        {   // de-interleave the records by consumer (tn_common.h: WalkHot / WalkTet / WalkFid)
            const size_t n4 = t->vars.n;
            const bool per_tet = t->writer_table ? t->writer_table == 2 : n4 / 4 >= tn::WALK_TET_MIN_TETS;
            t->hot.alloc(n4); t->fidt.alloc(n4);
            t->cold.release(); t->tets.release();
            if (per_tet) t->tets.alloc(n4 / 4); else t->cold.alloc(n4);
            tn::launch_split_walk_records(n4, t->vars.p, t->hot.p, per_tet ? nullptr : t->cold.p, per_tet ? t->tets.p : nullptr,
                                          t->fidt.p, stream);
            TN_HIP(hipStreamSynchronize(stream));
            t->vars.release();
        }
        m.hot = t->hot.p; m.cold = t->cold.n ? t->cold.p : nullptr; m.tets = t->tets.n ? t->tets.p : nullptr; m.fidt = t->fidt.p; m.n_hull = (uint32_t)n_hull;
        m.hull_nodes = reinterpret_cast<const float4 *>(t->hull_nodes.p);
        m.hull_tris = reinterpret_cast<const float4 *>(t->hull_tris.p);
        m.n_hull_nodes = (uint32_t)n_hull_nodes;
        t->loaded = true;
    });
}

size_t tn_num_faces(tn_tracer_t tracer) { return tracer && tracer->loaded ? tracer->mesh.F : 0; }

int tn_get_faces(tn_tracer_t tracer, uint32_t *faces_host, uint32_t *face_tets_host) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        if (!t->loaded) throw tn::Error("load_tetrahedra must be called first");
        if (t->host.face_tets.size() != 2 * (size_t)t->mesh.F) {   // device build: the tables live on the device only
            DeviceGuard g(t->device);
            t->host.faces.resize(3 * (size_t)t->mesh.F);
            t->host.face_tets.resize(2 * (size_t)t->mesh.F);
            if (t->mesh.F) {
                TN_HIP(hipMemcpy(t->host.faces.data(), t->faces.p, t->host.faces.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                TN_HIP(hipMemcpy(t->host.face_tets.data(), t->face_tets.p, t->host.face_tets.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            }
        }
        if (faces_host) std::memcpy(faces_host, t->host.faces.data(), t->host.faces.size() * sizeof(uint32_t));
        if (face_tets_host)
            std::memcpy(face_tets_host, t->host.face_tets.data(), t->host.face_tets.size() * sizeof(uint32_t));
    });
}

/* Test aid: copies one of the structures load_tetrahedra built to the host.  which: 0 faces, 1 face_tets, 2 walk records,
 * 3 hull nodes, 4 hull triangles, 5 BVH child rows, 6 BVH boxes, 7 BVH leaf ids, 8 BVH leaf triangles.  *bytes receives
 * the size; dst may be null (size query). */
int tn_get_build_table(tn_tracer_t tracer, int which, void *dst, size_t *bytes) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        if (!t->loaded) throw tn::Error("load_tetrahedra must be called first");
        DeviceGuard g(t->device);
        const void *src = nullptr;
        size_t n = 0;
        switch (which) {
            case 0: src = t->faces.p; n = t->faces.n * 4; break;
            case 1: src = t->face_tets.p; n = t->face_tets.n * 4; break;
            case 2: {   // the 64-byte records, re-assembled from the three tables (the unit of the build equality checks)
                const size_t n4 = t->hot.n;
                n = n4 * sizeof(tn::WalkVar);
                if (bytes) *bytes = n;
                if (dst && n4) {
                    std::vector<tn::WalkHot> h(n4); std::vector<tn::WalkTet> c(t->tets.n); std::vector<tn::WalkCold> cc(t->cold.n);
                    std::vector<tn::WalkFid> f(n4);
                    TN_HIP(hipMemcpy(h.data(), t->hot.p, n4 * sizeof(tn::WalkHot), hipMemcpyDeviceToHost));
                    if (!c.empty()) TN_HIP(hipMemcpy(c.data(), t->tets.p, c.size() * sizeof(tn::WalkTet), hipMemcpyDeviceToHost));
                    if (!cc.empty()) TN_HIP(hipMemcpy(cc.data(), t->cold.p, cc.size() * sizeof(tn::WalkCold), hipMemcpyDeviceToHost));
                    TN_HIP(hipMemcpy(f.data(), t->fidt.p, n4 * sizeof(tn::WalkFid), hipMemcpyDeviceToHost));
                    tn::WalkVar *o = static_cast<tn::WalkVar *>(dst);
                    for (size_t i = 0; i < n4; ++i) {
                        tn::WalkVar v{};
                        for (int k = 0; k < 3; ++k) v.pn[k] = h[i].pn[k];
                        v.nb[0] = h[i].nb0; v.nb[1] = h[i].nb1; v.nb[2] = h[i].nb2; v.code_lo = h[i].code_lo; v.code_hi = h[i].code_hi;
                        if (!c.empty()) { v.orig = c[i >> 2].orig; for (uint32_t k = 0; k < 4; ++k) v.vid[k] = c[i >> 2].vid((uint32_t)(i & 3), k); }
                        else { v.orig = cc[i].orig; for (int k = 0; k < 4; ++k) v.vid[k] = cc[i].vid[k]; }
                        v.fid0 = f[i].fid[0]; v.fid1 = f[i].fid[1]; v.fid2 = f[i].fid[2];
                        o[i] = v;
                    }
                }
                return;
            }
            case 3: src = t->hull_nodes.p; n = t->hull_nodes.n * 4; break;
            case 4: src = t->hull_tris.p; n = t->hull_tris.n * 4; break;
            case 5: src = t->bvh.child.p; n = t->bvh.child.n * 4; break;
            case 6: src = t->bvh.boxes.p; n = t->bvh.boxes.n * 4; break;
            case 7: src = t->bvh.leaf_id.p; n = t->bvh.leaf_id.n * 4; break;
            case 8: src = t->bvh.leaf_tri.p; n = t->bvh.leaf_tri.n * 4; break;
            default: throw tn::Error("unknown table");
        }
        if (bytes) *bytes = n;
        if (dst && n) TN_HIP(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    });
}

static tn::TraceParams make_params(tn_tracer *t, size_t R, uint32_t M, const float *o, const float *d,
                                   uint32_t *num, uint32_t *cells, float *bary, float *dist, uint32_t *verts) {
    tn::TraceParams p{};
    p.origins = o; p.dirs = d;
    p.faces = t->mesh.faces; p.face_tets = t->mesh.face_tets;
    p.bvh = t->mesh.bvh;
    p.out_num = num; p.out_cells = cells; p.out_bary = bary; p.out_dist = dist; p.out_verts = verts;
    p.M = M; p.num_items = R; p.ray_list = nullptr;
    p.stats = t->stats.p;
    return p;
}

static int trace_rays_common(tn_tracer_t tracer, size_t R, uint32_t M, const float *origins, const float *directions,
                             uint32_t *num_visited, uint32_t *visited, float *bary, float *dist, uint32_t *verts,
                             uint32_t flags, void *stream_) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        std::lock_guard<std::mutex> lock(t->mu);
        if (flags & ~(uint32_t)TN_TRACE_COMPACT_ROWS) throw tn::Error("unknown trace flag");
        // per CALL, not per tracer: a viewer thread and a trainer sharing one tracer may ask for different row forms
        const bool dense_tails = t->dense_tails && !(flags & TN_TRACE_COMPACT_ROWS);
        if (M == 0 || (M & (M - 1)) != 0) throw tn::Error("max_ray_triangles must be a power of 2.");
        if (!t->loaded) throw tn::Error("load_tetrahedra must be called first");
        if (M > 4096) throw tn::Error("max_ray_triangles larger than 4096 is not supported");
        if (R >= 0xFFFFFFFFull) throw tn::Error("too many rays for one call");
        if (R == 0) return;
        if (!origins || !directions || !num_visited || !visited || !bary || !dist)
            throw tn::Error("null ray / output pointer");
        DeviceGuard g(t->device);
        hipStream_t stream = (hipStream_t)stream_;
        t->last_stream = stream;
        t->last_num_rays = R;
        TN_HIP(hipMemsetAsync(t->stats.p, 0, (tn_tracer::N_STATS + tn_tracer::N_CTR) * sizeof(unsigned long long), stream));
        tn::TraceParams p = make_params(t, R, M, origins, directions, num_visited, visited, bary, dist, verts);
        p.compact_rows = dense_tails ? 0u : 1u;
        // Small batches are latency-bound: a lane walking ~180 dependent steps is slower than one
        // wavefront per ray through the wide BVH (measured: 4096 rays, 300k tets: 1.5 ms vs 0.75 ms),
        // so the walk is used from `walk_min_rays` on (use_walk == 2 forces it for any size).
        // M >= 4: the writer and the fills store 16-byte vectors into the rows; M is a power of two (checked above), so
        // from 4 on every row base is 16-byte aligned
        const size_t walk_min = !t->walk_min_auto ? t->walk_min_rays
                              : t->mesh.T >= 4000000u ? (size_t)6144 : t->mesh.T >= 2000000u ? (size_t)8192 : t->walk_min_rays;
        const bool walk = t->use_walk && (R >= walk_min || t->use_walk == 2) && M >= 4 &&
                          t->mesh.n_hull > 0 && t->mesh.n_hull < (1u << 24);   // (HullEntry keeps the face's slot in 24 bits)
        t->last_walk = walk;
        if (walk) {
            // main stream: walk (hits -> log; classes) -> segment writer -> tails [ceil32(n), K0) of the certified rows
            // `pre`:  tails [K0, M) of ALL rows, from the start of the call (speculative, see below)
            // `side`: literal pairing of the logged hits of the rays whose order the walk did not certify, beside the fill
            // `aux`:  BVH re-trace of the handful of fallback rays (one wavefront each, pure latency), from the walk on
            // Everything that writes rows is ordered behind the speculative fill, so a ray with more than K0 segments
            // (or a literal / fallback row) simply overwrites its slots.  The log holds 16 B per hit slot; calls whose log
            // would exceed the cap are processed in ray chunks (multiples of 4096 rays, the walk's XCD run), serially.
            if (t->fallback_list.n < R) { t->fallback_list.alloc(R); t->walk_n.alloc(R); t->literal_list.alloc(R); t->hull_entry.alloc(R); }
            const bool verify_risk = t->verify_risk && t->verify_stride;
            if (t->verify_stride) {
                // sized with the other scratch buffers, BEFORE the first launch of the call: an allocation in the middle of
                // the overlapped schedule would synchronise the device there (hipFree / hipMalloc).  (R entries: the blind
                // sample + the risk classes can, on a degenerate mesh, name every ray)
                if (t->verify_list.n < R) t->verify_list.alloc(R);
                if (verify_risk && t->risk_list.n < R) t->risk_list.alloc(R);
            }
            size_t cap_bytes = t->log_cap_bytes;
            if (!cap_bytes) {
                // the log lives for the tracer's lifetime: at most a quarter of what is free now (plus what it already
                // holds), at most 24 GB; a call that needs more runs in chunks instead of failing in hipMalloc
                size_t free_b = 0, total_b = 0;
                TN_HIP(hipMemGetInfo(&free_b, &total_b));
                cap_bytes = std::min<size_t>((free_b + t->hit_log.n * sizeof(uint4)) / 4, (size_t)24 << 30);
            }
            size_t chunk = cap_bytes / ((size_t)M * sizeof(uint4));
            chunk = chunk / 4096 * 4096;
            if (chunk < 4096) chunk = 4096;
            if (chunk > R) chunk = R;
            const size_t log_entries = (chunk + 255) / 256 * 256 * (size_t)M;
            if (t->hit_log.n < log_entries) t->hit_log.alloc(log_entries);
            const bool single = chunk >= R;
            auto chunk_params = [&](size_t base, size_t n) {
                tn::TraceParams q = make_params(t, n, M, origins + 3 * base, directions + 3 * base, num_visited + base, visited + base * M,
                                                bary + base * M * 6, dist + base * M * 2, verts ? verts + base * M * 4 : nullptr);
                q.compact_rows = dense_tails ? 0u : 1u;
                q.sort_passes = t->literal_sort_passes;
                return q;
            };
            size_t walk_reserve = 0;     // set by the one-chunk schedule when a speculative fill runs beside the walk
            auto launch_walk = [&](size_t base, size_t n) {
                tn::WalkParams w{};
                w.t = chunk_params(base, n);
                w.vars = t->mesh.hot;
                w.scene_max = t->mesh.bvh.scene_max;
                w.hull_nodes = t->mesh.hull_nodes;
                w.hull_tris = t->mesh.hull_tris;
                w.n_hull_nodes = t->mesh.n_hull_nodes;
                w.hull_flat = t->mesh.hull_nodes + 2 * (size_t)t->mesh.n_hull_nodes;
                w.n_hull_leaves = t->hull_flat ? tn::hull_flat_leaves(t->mesh.n_hull) : 0u;
                w.n_hull_groups = t->hull_flat ? tn::hull_flat_groups(t->mesh.n_hull) : 0u;
                w.n_hull = t->mesh.n_hull;
                w.hull_entry = t->hull_entry.p + base;
                w.fallback_list = t->fallback_list.p;
                w.fallback_count = t->fallback_count();
                w.literal_list = t->literal ? t->literal_list.p : nullptr;
                w.literal_count = t->literal_count();
                w.walk_n = t->walk_n.p + base;
                w.hit_log = t->hit_log.p;
                w.ray_base = base;
                w.risk_list = verify_risk ? t->risk_list.p : nullptr;
                w.risk_count = t->risk_count();
                w.risk_band = (float)t->risk_band;
                w.cert_ends = t->cert_ends == 2 ? (t->mesh.T >= tn::WALK_TET_MIN_TETS ? 1u : 3u) : (uint32_t)t->cert_ends;
                tn::launch_trace_walk(w, stream, walk_reserve);
                if (t->verify_stride && !single) {  // chunked call: serially, before anything that reads walk_n / the fallback list
                    tn::launch_verify_counts(w.t, t->verify_stride, w.walk_n, w.fallback_list, w.fallback_count, base, stream, false,
                                             t->verify_inject);
                    if (verify_risk)
                        tn::launch_verify_counts(w.t, t->verify_stride, w.walk_n, w.fallback_list, w.fallback_count, base, stream, false,
                                                 false, t->risk_list.p, t->risk_count(), n);
                }
            };
            auto launch_segments = [&](size_t base, size_t n) {
                tn::WriteParams q{};
                q.num_rays = n; q.M = M; q.dense_tails = dense_tails ? 1u : 0u;
                q.walk_n = t->walk_n.p + base;
                q.hit_log = t->hit_log.p;
                q.cold = t->mesh.cold; q.tets = t->mesh.tets;
                q.out_cells = visited + base * M;
                q.out_bary = bary + base * M * 6;
                q.out_dist = dist + base * M * 2;
                q.out_verts = verts ? verts + base * M * 4 : nullptr;
                tn::launch_write_segments(q, stream, t->writer_blocks);
            };
            auto launch_fill = [&](size_t base, size_t n, uint32_t k_hi, hipStream_t st) {   // [ceil32(n_r), k_hi) of the certified rows
                if (!dense_tails) return;
                tn::launch_fill_range(n, M, false, t->walk_n.p + base, num_visited + base, visited + base * M, bary + base * M * 6,
                                      dist + base * M * 2, verts ? verts + base * M * 4 : nullptr, st, k_hi, false, t->fill_blocks);
            };
            auto launch_literal = [&](size_t base, size_t n, hipStream_t st) {
                if (!t->literal) return;
                tn::launch_postprocess_log(chunk_params(base, n), t->mesh.fidt, t->hit_log.p, t->literal_list.p, t->literal_count(),
                                           n, st);
            };
            p.ray_list = t->fallback_list.p;
            p.item_count = t->fallback_count();
            if (single) {
                // Speculative tail fill: a ray of a uniform mesh of T tets crosses at most ~3.45 T^(1/3) faces, so
                // the slots from ceil32(3.6 T^(1/3)) + 32 on are constants in (almost) every row and can be streamed BESIDE
                // the walk (VALU-issue-bound, the fill HBM-write-bound).  The walk crawls beside a saturating write stream, so
                // only as many bytes as its own duration buys are filled that way: the last quarter of every row (measured:
                // profiles/r02p_specfill*.txt, r03a_sched.txt: +1..3 % per frame), and only where that quarter is mesh-safe
                // (round 3, non-resident fill: at 1M tets and M = 512, where rays reach 346 of the 384 slots, it cost 1..6 %; see below).
                uint32_t K0 = 0;
                if (t->spec_fill && dense_tails) {
                    K0 = (((uint32_t)(3.6 * std::cbrt((double)std::max<uint32_t>(t->mesh.T, 1u))) + 31u) & ~31u) + 32u;
                    const uint32_t quarter = (3u * M / 4u) & ~31u, half = (M / 2u) & ~31u;
                    // the longer the walk (the more faces per ray), the more bytes its duration hides: the last quarter of the
                    // rows on small meshes (C2: 384), the last half where rays reach beyond M/2 - 64 slots (C4: 256 measured
                    // best, 320 / 384: -3.6 / -0.6 % instead of -4.9 %); a row with more than K0 segments overwrites its slots
                    // meshes whose record tables the L2s no longer hold (the per-tet writer table's threshold): the walk waits
                    // for HBM itself and hides less -- the last quarter where the estimate (which carries a 32-slot margin) still
                    // allows it (C5, 1M tets: rays reach slot 346 of 384: -0.9 / -1.8 % in two runs, the last half +0.4 %,
                    // profiles/r04w_c5_specfill*.txt; round 3 had measured the non-resident fill slower there)
                    if (t->mesh.T >= tn::WALK_TET_MIN_TETS) K0 = K0 > quarter + 32u ? 0u : quarter;
                    else K0 = K0 > quarter ? 0u : (K0 + 32u > half ? half : quarter);
                    if (t->spec_k0) K0 = t->spec_k0 & ~31u;
                    if (K0 + 32u > M) K0 = 0;
                }
                // option "timing": the same kernels, serialised on the caller's stream with a timing event after each
                const bool timing = t->timing;
                hipStream_t s_pre = timing ? stream : t->pre, s_aux = timing ? stream : t->aux, s_side = timing ? stream : t->side;
                if (timing && !t->tev[0])
                    for (hipEvent_t &e : t->tev) TN_HIP(hipEventCreate(&e));
                int mark_i = 0;
                auto mark = [&] { if (timing) TN_HIP(hipEventRecord(t->tev[mark_i++], stream)); };
                mark();                                                   // 0: start
                // (round 6: [K0, M) of every row through k_fill_linear BEFORE the walk, the short [ceil32(n), K0) pieces after the
                // writer: +6 ... 15 % on the frames for K0 = 32 ... 192, +2 ... 8 % at 1M tets, profiles/r06ad_bulk_sweep*.txt; the same
                // with a block per (array, row): +11 ... 20 %, r06af_bulk_rows_sweep.txt)
                if (K0) {
                    TN_HIP(hipEventRecord(t->ev_start, stream));
                    TN_HIP(hipStreamWaitEvent(s_pre, t->ev_start, 0));
                    tn::launch_fill_range(R, M, true, t->walk_n.p, num_visited, visited, bary, dist, verts, s_pre, K0, true, t->spec_blocks);
                    TN_HIP(hipEventRecord(t->ev_pre, s_pre));
                    walk_reserve = timing ? 0 : (size_t)t->walk_lds_kb * 1024;
                }
                mark();                                                   // 1: speculative fill
                launch_walk(0, R);
                mark();                                                   // 2: walk
                TN_HIP(hipEventRecord(t->ev_fork, stream));
                TN_HIP(hipStreamWaitEvent(s_aux, t->ev_fork, 0));
                if (K0) {   // everything that writes rows comes after the speculative fill
                    TN_HIP(hipStreamWaitEvent(s_aux, t->ev_pre, 0));
                    TN_HIP(hipStreamWaitEvent(stream, t->ev_pre, 0));
                }
                tn::launch_trace_general(p, s_aux);
                mark();                                                   // 3: BVH re-trace of the fallback rays
                if (t->verify_stride) {
                    // the count cross-check beside the writer and the fill (late form): mismatching rays -> verify_list.  (Round 6
                    // measured two other places for it -- on the side stream behind the literal pairing, and on the aux stream but
                    // not before the writer has finished: +1.2..1.8 % on the frames, +0.2..3.5 % on C5, profiles/r06p_place_sweep.txt)
                    tn::launch_verify_counts(chunk_params(0, R), t->verify_stride, t->walk_n.p, t->verify_list.p, t->verify_count(), 0,
                                             s_aux, true, t->verify_inject);
                    // ... and EVERY certified ray of the risk classes (inside the wide band of a guard: DESIGN.md section 2)
                    if (verify_risk)
                        tn::launch_verify_counts(chunk_params(0, R), t->verify_stride, t->walk_n.p, t->verify_list.p, t->verify_count(), 0,
                                                 s_aux, true, false, t->risk_list.p, t->risk_count(), R);
                }
                mark();                                                   // 4: count cross-check
                TN_HIP(hipEventRecord(t->ev_aux, s_aux));
                // the segment writer is enqueued BEFORE the side stream's kernel: its grid is sized for the worst case (the
                // count lives on the device) and would otherwise take every wave slot first
                launch_segments(0, R);
                mark();                                                   // 5: segment writer
                TN_HIP(hipEventRecord(t->ev_seg, stream));
                TN_HIP(hipStreamWaitEvent(s_side, t->ev_seg, 0));    // literal pairing beside the bandwidth-bound fill, not
                launch_literal(0, R, s_side);                        // beside the latency-bound writer (r02f_sched_sweep.txt)
                mark();                                                   // 6: literal pairing of the logged hits
                // (round 6, once more: the tail fill needs only the walk's counts, but beside the writer it costs +3 ... 5 % whatever its
                // grid and whichever is enqueued first, profiles/r06w_fill_beside.txt)
                launch_fill(0, R, K0 ? K0 : M, stream);
                mark();                                                   // 7: tail fill
                TN_HIP(hipEventRecord(t->ev_join, s_side));
                TN_HIP(hipStreamWaitEvent(stream, t->ev_join, 0));
                TN_HIP(hipStreamWaitEvent(stream, t->ev_aux, 0));
                if (t->verify_stride) {
                    // rows of the rays whose count differed (none, as far as anyone has seen): whole rows, after every other
                    // writer of the call.  The count lives on the device: a small grid that finds it 0 and exits
                    tn::TraceParams pv = make_params(t, 64, M, origins, directions, num_visited, visited, bary, dist, verts);
                    pv.compact_rows = dense_tails ? 0u : 1u;
                    pv.ray_list = t->verify_list.p;
                    pv.item_count = t->verify_count();
                    tn::launch_trace_general(pv, stream);
                }
                mark();                                                   // 8: end
                t->tev_valid = timing;
            } else {
                for (size_t base = 0; base < R; base += chunk) {
                    const size_t n = R - base < chunk ? R - base : chunk;
                    TN_HIP(hipMemsetAsync(t->literal_count(), 0, sizeof(uint32_t), stream));
                    TN_HIP(hipMemsetAsync(t->risk_count(), 0, sizeof(uint32_t), stream));
                    launch_walk(base, n);
                    launch_segments(base, n);
                    launch_fill(base, n, M, stream);
                    launch_literal(base, n, stream);   // before the next chunk's walk reuses the log
                }
                tn::launch_trace_general(p, stream);
            }
        } else {
            // Small batch (below walk_min_rays): one wavefront per ray through the BVH.  Latency-bound, so every ray should
            // be resident at once: LDS hit arrays sized for the hits a ray of THIS mesh is expected to have (a uniform mesh
            // of T tets: at most ~3.45 T^(1/3) faces on a ray), rays with more go through a second launch
            // with the full M-entry arrays.
            uint32_t C = 64;
            const double expect = 3.6 * std::cbrt((double)std::max<uint32_t>(t->mesh.T, 1u));
            while (C < expect && C < M) C <<= 1;
            if (t->lds_cap) C = t->lds_cap;
            if (t->small_lds && C < M) {
                if (t->fallback_list.n < R) { t->fallback_list.alloc(R); t->walk_n.alloc(R); t->literal_list.alloc(R); t->hull_entry.alloc(R); }
                tn::TraceParams p1 = p;
                p1.lds_cap = C; p1.overflow_list = t->fallback_list.p; p1.overflow_count = t->fallback_count();
                tn::launch_trace_general(p1, stream);
                p.ray_list = t->fallback_list.p;
                p.item_count = t->fallback_count();
            }
            tn::launch_trace_general(p, stream);
        }
        TN_HIP(hipGetLastError());
    });
}

int tn_trace_rays(tn_tracer_t tracer, size_t R, uint32_t M, const float *origins, const float *directions,
                  uint32_t *num_visited, uint32_t *visited, float *bary, float *dist, uint32_t *verts,
                  void *stream_) {
    return trace_rays_common(tracer, R, M, origins, directions, num_visited, visited, bary, dist, verts, 0u, stream_);
}

int tn_trace_rays_ex(tn_tracer_t tracer, size_t R, uint32_t M, const float *origins, const float *directions,
                     uint32_t *num_visited, uint32_t *visited, float *bary, float *dist, uint32_t *verts,
                     uint32_t flags, void *stream_) {
    return trace_rays_common(tracer, R, M, origins, directions, num_visited, visited, bary, dist, verts, flags, stream_);
}

int tn_find_matched_cells_indexed(size_t R, size_t S, size_t M, const uint32_t *ray_index, const uint32_t *num_visited,
                                  const uint32_t *visited, const float *dist, const float *bary, const float *distances,
                                  const uint32_t *verts, uint32_t *cells_out, uint32_t *verts_out, uint8_t *mask_out,
                                  float *bary_out, const uint32_t *count, void *stream_) {
    return guarded([&] {
        if (R == 0 || S == 0) return;
        if (!ray_index) throw tn::Error("ray_index is null");
        if (S >= 0xFFFFFFFFull || M >= 0xFFFFFFFFull) throw tn::Error("num_samples / max_visited_cells too large");
        if (M * 2 * sizeof(float) > 64 * 1024) throw tn::Error("max_visited_cells larger than 8192 is not supported");
        tn::launch_find_matched_cells(R, S, M, num_visited, visited, dist, bary, distances, verts, cells_out,
                                      verts_out, mask_out, bary_out, (hipStream_t)stream_, ray_index, count);
        TN_HIP(hipGetLastError());
    });
}

int tn_postprocess_hits(tn_tracer_t tracer, size_t R, uint32_t M, const uint32_t *hit_count,
                        const uint32_t *hit_ids, const float *hit_t, const float *hit_uv,
                        uint32_t *num_visited, uint32_t *visited, float *bary, float *dist, uint32_t *verts,
                        void *stream_) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        if (M == 0 || (M & (M - 1)) != 0) throw tn::Error("max_ray_triangles must be a power of 2.");
        if (!t->loaded) throw tn::Error("load_tetrahedra must be called first");
        if (R == 0) return;
        DeviceGuard g(t->device);
        hipStream_t stream = (hipStream_t)stream_;
        tn::TraceParams p = make_params(t, R, M, nullptr, nullptr, num_visited, visited, bary, dist, verts);
        p.stats = nullptr;
        tn::launch_postprocess_hits(p, hit_count, hit_ids, hit_t, hit_uv, stream);
        TN_HIP(hipGetLastError());
    });
}

int tn_postprocess_hits_tables(int device, size_t R, uint32_t M, const uint32_t *faces, const uint32_t *face_tets,
                               const uint32_t *hit_count, const uint32_t *hit_ids, const float *hit_t,
                               const float *hit_uv, uint32_t *num_visited, uint32_t *visited, float *bary, float *dist,
                               uint32_t *verts, void *stream_) {
    return guarded([&] {
        if (M == 0 || (M & (M - 1)) != 0) throw tn::Error("max_ray_triangles must be a power of 2.");
        if (!faces || !face_tets) throw tn::Error("null face table");
        if (R == 0) return;
        DeviceGuard g(device);
        tn::TraceParams p{};
        p.M = M;
        p.num_items = R;
        p.faces = faces;
        p.face_tets = face_tets;
        p.out_num = num_visited;
        p.out_cells = visited;
        p.out_bary = bary;
        p.out_dist = dist;
        p.out_verts = verts;
        tn::launch_postprocess_hits(p, hit_count, hit_ids, hit_t, hit_uv, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_trace_rays_triangles(tn_tracer_t tracer, size_t R, uint32_t M, const float *origins, const float *directions,
                            uint32_t *num_visited, uint32_t *visited, float *bary, float *dist, uint32_t *verts,
                            void *stream_) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        if (M == 0 || (M & (M - 1)) != 0) throw tn::Error("max_ray_triangles must be a power of 2.");
        if (!t->loaded) throw tn::Error("load_tetrahedra must be called first");
        if (M > 4096) throw tn::Error("max_ray_triangles larger than 4096 is not supported");
        if (R == 0) return;
        DeviceGuard g(t->device);
        tn::TraceParams p = make_params(t, R, M, origins, directions, num_visited, nullptr, nullptr, nullptr, nullptr);
        p.stats = nullptr;
        tn::launch_trace_triangles(p, visited, dist, bary, verts, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_find_tetrahedra(tn_tracer_t tracer, size_t N, const float *positions, uint32_t *tetrahedra, float *bary,
                       uint32_t *verts, void *stream_) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        if (!t->loaded) throw tn::Error("load_tetrahedra must be called first");
        if (N == 0) return;
        DeviceGuard g(t->device);
        tn::TraceParams p = make_params(t, N, 512, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        p.stats = nullptr;
        tn::launch_find_tetrahedra(p, positions, tetrahedra, bary, verts, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_trace_stats(tn_tracer_t tracer, uint64_t stats[4]) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        std::lock_guard<std::mutex> lock(t->mu);
        DeviceGuard g(t->device);
        TN_HIP(hipStreamSynchronize(t->last_stream));
        unsigned long long h[24];
        TN_HIP(hipMemcpy(h, t->stats.p, sizeof h, hipMemcpyDeviceToHost));
        for (int i = 0; i < 4; ++i) stats[i] = h[i];
        if (t->last_walk) {
            // not certified by the walk: literal pairing of the logged hits (h[4 + 13]) + BVH re-trace
            uint32_t fb = 0;
            TN_HIP(hipMemcpy(&fb, t->fallback_count(), sizeof fb, hipMemcpyDeviceToHost));
            fb += (uint32_t)h[4 + 13];
            stats[1] = fb;
            stats[0] = t->last_num_rays - fb;
        } else {
            stats[0] = 0;
            stats[1] = t->last_num_rays;
        }
    });
}

int tn_trace_flag_reasons(tn_tracer_t tracer, uint64_t reasons[16]) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        std::lock_guard<std::mutex> lock(t->mu);
        DeviceGuard g(t->device);
        TN_HIP(hipStreamSynchronize(t->last_stream));
        unsigned long long h[20];
        TN_HIP(hipMemcpy(h, t->stats.p, sizeof h, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i) reasons[i] = h[4 + i];
    });
}

int tn_trace_cross_check(tn_tracer_t tracer, uint64_t out[8]) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        std::lock_guard<std::mutex> lock(t->mu);
        if (!out) throw tn::Error("out is null");
        DeviceGuard g(t->device);
        TN_HIP(hipStreamSynchronize(t->last_stream));
        unsigned long long h[tn_tracer::N_STATS];
        TN_HIP(hipMemcpy(h, t->stats.p, sizeof h, hipMemcpyDeviceToHost));
        out[0] = t->verify_stride; out[1] = h[4 + 15]; out[2] = h[4 + 14];
        out[3] = h[24]; out[4] = h[25]; out[5] = h[26]; out[6] = h[27]; out[7] = 0;
    });
}

int tn_trace_timings(tn_tracer_t tracer, float ms[8]) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        std::lock_guard<std::mutex> lock(t->mu);
        if (!ms) throw tn::Error("ms is null");
        if (!t->tev_valid) throw tn::Error("no timed call: set option \"timing\" = 1 and trace a one-chunk walk call first");
        DeviceGuard g(t->device);
        TN_HIP(hipEventSynchronize(t->tev[tn_tracer::N_TEV - 1]));
        for (int i = 0; i + 1 < tn_tracer::N_TEV; ++i) TN_HIP(hipEventElapsedTime(&ms[i], t->tev[i], t->tev[i + 1]));
    });
}

int tn_fill_rows(size_t R, uint32_t M, uint32_t first_slot, uint32_t *visited, float *bary, float *dist, uint32_t *verts,
                 void *stream_) {
    return guarded([&] {
        if (R == 0) return;
        if (!visited || !bary || !dist) throw tn::Error("null output pointer");
        if (M < 4 || (M & (M - 1)) != 0) throw tn::Error("max_ray_triangles must be a power of 2.");
        if (first_slot >= M) return;
        // rows are written from a 128-byte line boundary of all four arrays on (multiples of 32 slots), like the tracer's own fill;
        // any other first slot is refused rather than rounded: rounding down would overwrite up to 31 written segments
        if (first_slot & 31u) throw tn::Error("tn_fill_rows: first_slot must be a multiple of 32");
        tn::launch_fill_range(R, M, true, nullptr, nullptr, visited, bary, dist, verts, (hipStream_t)stream_, first_slot, false, tn::FILL_LINEAR);   // no per-row lookups here: one linear stream per array
        TN_HIP(hipGetLastError());
    });
}

int tn_set_option(tn_tracer_t tracer, const char *name, int value) {
    return guarded([&] {
        tn_tracer *t = checked(tracer);
        std::lock_guard<std::mutex> lock(t->mu);
        const std::string k = name ? name : "";
        if (k == "gpu_build") t->gpu_build = value != 0;
        else if (k == "timing") { t->timing = value != 0; t->tev_valid = false; }
        else if (k == "leaf_width") {
            if (value != 16 && value != 32 && value != 64) throw tn::Error("leaf_width must be 16, 32 or 64");
            t->leaf_width = (unsigned)value;
        }
        else if (k == "walk") t->use_walk = value < 0 ? 0 : (value > 2 ? 2 : value);
        else if (k == "walk_min_rays") { t->walk_min_rays = value < 0 ? 0 : (size_t)value; t->walk_min_auto = false; }
        else if (k == "dense_tails") t->dense_tails = value != 0;
        else if (k == "literal") t->literal = value != 0;
        else if (k == "spec_fill") t->spec_fill = value != 0;
        else if (k == "spec_k0") t->spec_k0 = (unsigned)value;
        else if (k == "spec_blocks") t->spec_blocks = value == -2 ? tn::FILL_LINEAR : value < 0 ? tn::FILL_FINE : (unsigned)value;
        else if (k == "hull_flat") t->hull_flat = value != 0;
        else if (k == "writer_blocks") t->writer_blocks = (unsigned)value;
        else if (k == "fill_blocks") t->fill_blocks = value == -2 ? tn::FILL_LINEAR : value < 0 ? tn::FILL_FINE : (unsigned)value;    // -1: one block per row
        else if (k == "walk_lds_kb") t->walk_lds_kb = (unsigned)value;
        else if (k == "small_lds") t->small_lds = value != 0;
        else if (k == "lds_cap") {
            if (value < 0 || (value & (value - 1)) != 0 || (value && value < 8)) throw tn::Error("lds_cap must be 0 or a power of two >= 8");
            t->lds_cap = (unsigned)value;
        }
        else if (k == "writer_table") t->writer_table = value;   // applies at the next load_tetrahedra
        else if (k == "cert_ends") { if (value < 0 || value > 3) throw tn::Error("cert_ends must be 0 .. 3"); t->cert_ends = value; }
        else if (k == "verify_inject") t->verify_inject = value != 0;
        else if (k == "literal_sort_passes") t->literal_sort_passes = value < 0 ? 0u : (unsigned)value;
        else if (k == "verify_stride") t->verify_stride = value < 0 ? 0u : (unsigned)value;
        else if (k == "verify_risk") t->verify_risk = value != 0;
        else if (k == "risk_band") t->risk_band = value < 1 ? 1u : (unsigned)value;
        else if (k == "log_cap_mb") t->log_cap_bytes = value <= 0 ? 0 : (size_t)value << 20;
        else throw tn::Error("unknown option " + (name ? k : std::string("(null)")));
    });
}

int tn_find_matched_cells(size_t R, size_t S, size_t M, const uint32_t *num_visited, const uint32_t *visited,
                          const float *dist, const float *bary, const float *distances, const uint32_t *verts,
                          uint32_t *cells_out, uint32_t *verts_out, uint8_t *mask_out, float *bary_out,
                          void *stream_) {
    return guarded([&] {
        if (R == 0 || S == 0) return;
        if (S >= 0xFFFFFFFFull || M >= 0xFFFFFFFFull) throw tn::Error("num_samples / max_visited_cells too large");
        if (M * 2 * sizeof(float) > 64 * 1024) throw tn::Error("max_visited_cells larger than 8192 is not supported");
        tn::launch_find_matched_cells(R, S, M, num_visited, visited, dist, bary, distances, verts, cells_out,
                                      verts_out, mask_out, bary_out, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_interpolate_values(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                          const float *field, float *result, void *stream_) {
    return guarded([&] {
        tn::launch_interpolate_values(D, V, n, Fd, vi, bc, field, result, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_interpolate_values_backward(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi,
                                   const float *bc, const float *grad_in, float *field_grad_out, void *stream_) {
    return guarded([&] {
        tn::launch_interpolate_values_backward(D, V, n, Fd, vi, bc, grad_in, false, field_grad_out, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_interpolate_values_backward_rows(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi,
                                        const float *bc, const float *grad_rows, float *field_grad_out, void *stream_) {
    return guarded([&] {
        tn::launch_interpolate_values_backward(D, V, n, Fd, vi, bc, grad_rows, true, field_grad_out, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

/* vertex-major variants: the caller keeps a [V, Fd] shadow of the field (tn_transpose_f32 makes it, once per field
 * version) -- no per-call O(V) transposition, no temporaries */
int tn_transpose_f32(uint32_t rows, uint32_t cols, const float *in, float *out, void *stream_) {
    return guarded([&] {
        tn::launch_transpose(in, out, rows, cols, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_interpolate_values_vm(uint32_t D, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                             const float *field_vm, float *result, void *stream_) {
    return guarded([&] {
        tn::launch_interpolate_values_vm(D, n, Fd, vi, bc, field_vm, result, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_interpolate_values_backward_vm(uint32_t D, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                                      const float *grad_rows, float *field_grad_vm, void *stream_) {
    return guarded([&] {
        tn::launch_interpolate_values_backward_vm(D, n, Fd, vi, bc, grad_rows, field_grad_vm, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_interpolate_values_backward_vm_det(uint32_t D, uint32_t V, uint32_t n, uint32_t Fd, const uint32_t *vi, const float *bc,
                                          const float *grad_rows, float *field_grad_vm, void *stream_) {
    return guarded([&] {
        if (n == 0 || Fd == 0) return;
        if (!vi || !bc || !grad_rows || !field_grad_vm) throw tn::Error("null pointer");
        if (V == 0) throw tn::Error("interpolate_values backward (deterministic): the vertex count is 0");
        // vertex ids >= V (TN_EMPTY = an unmatched slot, and any other out-of-range id) sort behind every vertex's run and are
        // skipped; the atomic entry point would write through such an id (the reference does not check either, py_binding.cpp:309-311)
        tn::launch_interpolate_values_backward_vm_det(D, V, n, Fd, vi, bc, grad_rows, field_grad_vm, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

/* ---- shallow MLP: a handle owns the packed forms of one set of weights and the per-call scratch ---- */
}  // extern "C" (the handle type is C++)

struct tn_mlp {
    int device = 0;
    tn::DevBuf<float> pk_plain, pk_gather, pt, enc, grad_scratch, wenc, hterm;
    tn::DevBuf<uint4> blob;
    tn::DevBuf<float> render_scratch;    // per-block hand-over area of tn_render_rays (grown on demand, never shrunk)
    tn::DevBuf<unsigned long long> render_prof;   // TETRANERF_HIP_RENDER_PROFILE=1 (debug): phase ticks of tn_render_rays
    bool packed = false;
    // per-call scratch: grown on demand (blocking hipMalloc, rare), never shrunk; one handle serves one stream at a time
    tn::MlpPacks packs(size_t rays) {
        if (!packed) throw tn::Error("tn_mlp_set_weights must be called first");
        if (enc.n < rays * tn::mlp_enc_floats_per_ray() || hterm.n < rays * 128) {
            const size_t cap = std::max<size_t>(rays + rays / 4, 4096);
            TN_HIP(hipDeviceSynchronize());   // the old scratch may still be in use by queued kernels
            enc.alloc(cap * tn::mlp_enc_floats_per_ray());
            hterm.alloc(cap * 128);
        }
        return tn::MlpPacks{pk_plain.p, pk_gather.p, pt.p, blob.p, wenc.p, hterm.p, enc.p, nullptr, grad_scratch.p};
    }
};

namespace {
// NULL = the reference configuration's default: white, training-mode renderer (no clamp)
tn::Background background_of(const tn_rgb_background *b) {
    return b ? tn::Background{b->r, b->g, b->b, b->clamp} : tn::Background{1.f, 1.f, 1.f, 0};
}
tn_mlp *checked_mlp(tn_mlp_t m) {
    if (!m) throw tn::Error("mlp handle is null");
    return m;
}
void check_mode(int mode) {
    if (mode != 0 && mode != 1) throw tn::Error("mlp mode must be 0 (fp32 MFMA) or 1 (bf16x3 MFMA)");
}
}  // namespace

extern "C" {

int tn_mlp_create(int device, tn_mlp_t *out) {
    return guarded([&] {
        if (!out) throw tn::Error("out is null");
        int count = 0;
        TN_HIP(hipGetDeviceCount(&count));
        if (device < 0 || device >= count) throw tn::Error("The device argument must be a CUDA device.");
        DeviceGuard g(device);
        auto m = std::make_unique<tn_mlp>();
        m->device = device;
        m->pk_plain.alloc(tn::mlp_pack_floats());
        m->pk_gather.alloc(tn::mlp_pack_floats());
        m->pt.alloc(tn::mlp_backward_pack_floats());
        m->blob.alloc(tn::mlp_x3_blob_u4());
        m->wenc.alloc(128 * 28);
        *out = m.release();
    });
}

int tn_mlp_destroy(tn_mlp_t mlp) {
    return guarded([&] {
        if (!mlp) return;
        DeviceGuard g(mlp->device);
        (void)hipDeviceSynchronize();
        delete mlp;
    });
}

int tn_mlp_set_weights(tn_mlp_t mlp, const tn_mlp_weights *w, void *stream_) {
    return guarded([&] {
        tn_mlp *m = checked_mlp(mlp);
        if (!w) throw tn::Error("null pointer");
        const float *const all[12] = {w->w1, w->b1, w->w2, w->b2, w->w3, w->b3, w->wd, w->bd, w->wh, w->bh, w->wr, w->br};
        for (const float *x : all) if (!x) throw tn::Error("null weight pointer");
        DeviceGuard g(m->device);
        hipStream_t stream = (hipStream_t)stream_;
        tn::MlpWeights mw{w->w1, w->b1, w->w2, w->b2, w->w3, w->b3, w->wd, w->bd, w->wh, w->bh, w->wr, w->br};
        tn::launch_mlp_pack(mw, m->pk_plain.p, false, stream);
        tn::launch_mlp_pack(mw, m->pk_gather.p, true, stream);
        tn::launch_mlp_pack_t(mw, m->pt.p, stream);
        tn::launch_mlp_pack_x3(mw, m->blob.p, stream);
        tn::launch_pack_wenc(mw, m->wenc.p, stream);
        TN_HIP(hipGetLastError());
        m->packed = true;
    });
}

int tn_mlp_forward(tn_mlp_t mlp, size_t n, uint32_t samples_per_ray, const float *feats, const float *dirs, int mode,
                   float *sigma, float *rgb, void *stream_) {
    return guarded([&] {
        tn_mlp *m = checked_mlp(mlp);
        check_mode(mode);
        if (n == 0) return;
        if (!feats || !sigma || (rgb && !dirs)) throw tn::Error("null pointer");
        if (samples_per_ray == 0 || n % samples_per_ray != 0) throw tn::Error("n must be a multiple of samples_per_ray");
        DeviceGuard g(m->device);
        const size_t rays = n / samples_per_ray;
        (mode ? tn::launch_mlp_forward_x3 : tn::launch_mlp_forward)(
            n, samples_per_ray, rays, feats, nullptr, nullptr, nullptr, dirs, m->packs(rays), sigma, rgb, (hipStream_t)stream_, nullptr);
        TN_HIP(hipGetLastError());
    });
}

int tn_mlp_forward_gather(tn_mlp_t mlp, size_t n, uint32_t samples_per_ray, const uint32_t *vertex_indices,
                          const float *barycentric, const float *field_vm, const float *dirs, int mode, float *sigma,
                          float *rgb, const float *ray_head_bias, const uint32_t *count, void *stream_) {
    return guarded([&] {
        tn_mlp *m = checked_mlp(mlp);
        check_mode(mode);
        if (n == 0) return;
        if (!vertex_indices || !barycentric || !field_vm || !sigma || (rgb && !dirs)) throw tn::Error("null pointer");
        if (samples_per_ray == 0 || n % samples_per_ray != 0) throw tn::Error("n must be a multiple of samples_per_ray");
        DeviceGuard g(m->device);
        const size_t rays = n / samples_per_ray;
        tn::MlpPacks pk = m->packs(rays);
        pk.ray_bias = rgb ? ray_head_bias : nullptr;
        (mode ? tn::launch_mlp_forward_x3 : tn::launch_mlp_forward)(
            n, samples_per_ray, rays, nullptr, vertex_indices, barycentric, field_vm, dirs, pk, sigma, rgb,
            (hipStream_t)stream_, count);
        TN_HIP(hipGetLastError());
    });
}

int tn_render_rays(tn_mlp_t mlp, uint32_t M, const uint32_t *num_visited, const float *hit_distances, const float *barycentric,
                   const uint32_t *vertex_indices, const uint32_t *ray_index, const uint32_t *count, size_t num_hit_rays_max,
                   uint32_t num_samples, uint32_t num_fine, int biased, const float *linspace, const float *u_table,
                   float histogram_padding, float eps, const float *field_vm, const float *dirs,
                   const tn_rgb_background *background, float *out_rgb, float *out_acc, float *out_depth,
                   const float *ray_head_bias, void *stream_) {
    return tn_render_rays_ex(mlp, M, num_visited, hit_distances, barycentric, vertex_indices, ray_index, count, num_hit_rays_max,
                             num_samples, num_fine, biased, linspace, u_table, histogram_padding, eps, field_vm, dirs, background,
                             out_rgb, out_acc, out_depth, ray_head_bias, 0, stream_);
}

int tn_render_rays_ex(tn_mlp_t mlp, uint32_t M, const uint32_t *num_visited, const float *hit_distances, const float *barycentric,
                      const uint32_t *vertex_indices, const uint32_t *ray_index, const uint32_t *count, size_t num_hit_rays_max,
                      uint32_t num_samples, uint32_t num_fine, int biased, const float *linspace, const float *u_table,
                      float histogram_padding, float eps, const float *field_vm, const float *dirs,
                      const tn_rgb_background *background, float *out_rgb, float *out_acc, float *out_depth,
                      const float *ray_head_bias, int mode, void *stream_) {
    return guarded([&] {
        tn_mlp *m = checked_mlp(mlp);
        check_mode(mode);
        if (num_hit_rays_max == 0) return;
        if (!num_visited || !hit_distances || !barycentric || !vertex_indices || !ray_index || !linspace || !field_vm || !dirs ||
            !out_rgb || !out_acc || !out_depth || (num_fine && !u_table))
            throw tn::Error("null pointer");
        if (num_samples == 0) throw tn::Error("num_samples must be positive");
        if (num_hit_rays_max >= 0xFFFFFFFFull) throw tn::Error("too many rays for one call");
        if ((size_t)num_samples + num_fine + 2 > 8192) throw tn::Error("render_rays: too many samples per ray");
        DeviceGuard g(m->device);
        if (!m->packed) throw tn::Error("tn_mlp_set_weights must be called first");
        const unsigned grid = 256;   // one persistent 8-wave block per CU (tn_render_rays.hip)
        tn::RenderRaysLayout L{};
        const size_t need = tn::render_rays_scratch_floats(num_hit_rays_max, num_samples, num_fine, ray_head_bias != nullptr, grid, L);
        if (m->render_scratch.n < need) {
            TN_HIP(hipDeviceSynchronize());   // the old scratch may still be in use by queued kernels
            m->render_scratch.alloc(need + need / 8);
        }
        // debug aid: TETRANERF_HIP_RENDER_PROFILE=1 prints where the persistent kernel's blocks spent their time, per call
        // (a stream synchronisation per call: for profiling runs only)
#if defined(TN_RENDER_DIAG) && TN_RENDER_DIAG
        static const bool profile = env_flag("TETRANERF_HIP_RENDER_PROFILE", false);   // diagnostic builds only (tn_render_rays.hip)
#else
        constexpr bool profile = false;
#endif
        if (profile) {
            if (!m->render_prof.p) m->render_prof.alloc(8);
            TN_HIP(hipMemsetAsync(m->render_prof.p, 0, 8 * sizeof(unsigned long long), (hipStream_t)stream_));
        }
        tn::launch_render_rays(num_visited, hit_distances, barycentric, vertex_indices, M, ray_index, count, num_hit_rays_max, num_samples,
                               num_fine, biased != 0, linspace, u_table, histogram_padding, eps, field_vm, dirs, ray_head_bias, m->packs(0),
                               background_of(background), out_rgb, out_acc, out_depth, m->render_scratch.p, L, grid, (hipStream_t)stream_,
                               profile ? m->render_prof.p : nullptr, mode);
        TN_HIP(hipGetLastError());
        if (profile) {
            unsigned long long h[8];
            TN_HIP(hipStreamSynchronize((hipStream_t)stream_));
            TN_HIP(hipMemcpy(h, m->render_prof.p, sizeof h, hipMemcpyDeviceToHost));
            const double nb = h[5] ? (double)h[5] : 1.0, us = 0.01;   // 100 MHz ticks -> microseconds, mean per working block
            fprintf(stderr, "[tn_render_rays] S=%u fine=%u rays<=%zu blocks=%llu  mean us per block: sample+match %.1f | mlp density %.1f | "
                            "weights+pdf+match %.1f | mlp full %.1f | composite %.1f\n", num_samples, num_fine, num_hit_rays_max,
                    h[5], h[0] * us / nb, h[1] * us / nb, h[2] * us / nb, h[3] * us / nb, h[4] * us / nb);
        }
    });
}

namespace {
tn::MlpBackwardBuffers training_buffers(const tn_mlp_backward_buffers *b) {
    return tn::MlpBackwardBuffers{b->x0, b->h1, b->h2, b->h3, b->h4, (unsigned long long *)b->masks,
                                  b->d1, b->d2, b->d3, b->d4, b->dhead, b->dx0};
}
}  // namespace

int tn_mlp_forward_gather_train(tn_mlp_t mlp, size_t n, uint32_t samples_per_ray, const uint32_t *vertex_indices,
                                const float *barycentric, const float *field_vm, const float *dirs, float *sigma, float *rgb,
                                const tn_mlp_backward_buffers *b, const float *ray_head_bias, void *stream_) {
    return guarded([&] {
        tn_mlp *m = checked_mlp(mlp);
        if (n == 0) return;
        if (!vertex_indices || !barycentric || !field_vm || !dirs || !sigma || !rgb || !b) throw tn::Error("null pointer");
        if (!b->x0 || !b->h1 || !b->h2 || !b->h3 || !b->h4 || !b->masks) throw tn::Error("null pointer");
        if (samples_per_ray == 0 || n % samples_per_ray != 0) throw tn::Error("n must be a multiple of samples_per_ray");
        DeviceGuard g(m->device);
        const size_t rays = n / samples_per_ray;
        tn::MlpPacks pk = m->packs(rays);
        pk.ray_bias = ray_head_bias;
        tn::launch_mlp_forward_train(n, samples_per_ray, rays, vertex_indices, barycentric, field_vm, dirs, pk, sigma, rgb,
                                     training_buffers(b), (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_mlp_backward(tn_mlp_t mlp, size_t n, const float *sigma, const float *rgb, const float *d_sigma, const float *d_rgb,
                    const tn_mlp_backward_buffers *b, void *stream_) {
    return guarded([&] {
        tn_mlp *m = checked_mlp(mlp);
        if (n == 0) return;
        if (!b || !sigma || !rgb || !d_sigma || !d_rgb) throw tn::Error("null pointer");
        if (!b->masks || !b->d1 || !b->d2 || !b->d3 || !b->d4 || !b->dhead || !b->dx0) throw tn::Error("null pointer");
        DeviceGuard g(m->device);
        tn::launch_mlp_backward(n, sigma, rgb, m->packs(0), d_sigma, d_rgb, training_buffers(b), (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_mlp_ray_head_grad(size_t n, uint32_t samples_per_ray, const tn_mlp_backward_buffers *b, float *d_ray_head_bias,
                         void *stream_) {
    return guarded([&] {
        if (n == 0) return;
        if (!b || !b->d4 || !d_ray_head_bias) throw tn::Error("null pointer");
        if (samples_per_ray == 0 || n % samples_per_ray != 0) throw tn::Error("n must be a multiple of samples_per_ray");
        tn::launch_ray_head_grad(n, samples_per_ray, b->d4, d_ray_head_bias, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_mlp_param_grads(tn_mlp_t mlp, size_t n, uint32_t samples_per_ray, const float *dirs, const tn_mlp_backward_buffers *b,
                       const tn_mlp_grads *grads, void *stream_) {
    return guarded([&] {
        tn_mlp *m = checked_mlp(mlp);
        if (n == 0) return;
        if (!b || !grads || !dirs) throw tn::Error("null pointer");
        if (samples_per_ray == 0 || n % samples_per_ray != 0) throw tn::Error("n must be a multiple of samples_per_ray");
        float *const gp[12] = {grads->w1, grads->b1, grads->w2, grads->b2, grads->w3, grads->b3,
                               grads->wd, grads->bd, grads->wh, grads->bh, grads->wr, grads->br};
        for (float *p : gp)
            if (!p) throw tn::Error("null pointer");
        if (!b->x0 || !b->h1 || !b->h2 || !b->h3 || !b->h4 || !b->d1 || !b->d2 || !b->d3 || !b->d4 || !b->dhead)
            throw tn::Error("null pointer");
        DeviceGuard g(m->device);
        if (!m->grad_scratch.p) {   // first training call of this handle
            TN_HIP(hipDeviceSynchronize());
            m->grad_scratch.alloc(tn::mlp_param_grad_scratch_floats());
        }
        const tn::MlpBackwardBuffers bb = training_buffers(b);
        tn::MlpParamGrads pg{gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], gp[6], gp[7], gp[8], gp[9], gp[10], gp[11]};
        tn::launch_mlp_param_grads(n, samples_per_ray, dirs, m->packs(n / samples_per_ray), bb, pg, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_compact_hits(size_t num_rays, const uint32_t *num_visited, uint32_t *order, uint32_t *count, uint32_t *padded,
                    uint32_t *scratch, size_t scratch_len, void *stream_) {
    return guarded([&] {
        if (!num_visited || !order || !count || !scratch) throw tn::Error("null pointer");
        if (num_rays >= 0xFFFFFFFFull) throw tn::Error("too many rays for one call");
        if (scratch_len < tn::compact_scratch_u32(num_rays)) throw tn::Error("compact_hits: scratch too small (2 * ceil(num_rays / 2048) uint32)");
        tn::launch_compact_hits(num_rays, num_visited, order, count, padded, scratch, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_sample_coarse(size_t num_hit_rays, uint32_t num_samples, uint32_t M, const uint32_t *ray_index, const uint32_t *num_visited,
                     const float *hit_distances, const float *linspace, const float *t_rand, int biased, float *edges,
                     float *near_far, const uint32_t *count, void *stream_) {
    return guarded([&] {
        if (num_hit_rays == 0) return;
        if (!ray_index || !num_visited || !hit_distances || !linspace || !edges || !near_far) throw tn::Error("null pointer");
        if (num_samples == 0) throw tn::Error("num_samples must be positive");
        tn::launch_sample_coarse(num_hit_rays, num_samples, M, ray_index, num_visited, hit_distances, linspace, t_rand, biased != 0,
                                 edges, near_far, (hipStream_t)stream_, count);
        TN_HIP(hipGetLastError());
    });
}

int tn_sample_pdf(size_t num_hit_rays, uint32_t num_samples, uint32_t num_fine, const float *edges, const float *weights,
                  const float *near_far, const float *u_table, const float *u_rand, float histogram_padding, float eps,
                  float *edges_out, const uint32_t *count, void *stream_) {
    return guarded([&] {
        if (num_hit_rays == 0) return;
        if (!edges || !weights || !near_far || !u_table || !edges_out) throw tn::Error("null pointer");
        if (num_samples == 0) throw tn::Error("num_samples must be positive");
        tn::launch_sample_pdf(num_hit_rays, num_samples, num_fine, edges, weights, near_far, u_table, u_rand, histogram_padding, eps,
                              edges_out, (hipStream_t)stream_, count);
        TN_HIP(hipGetLastError());
    });
}

int tn_composite_backward(size_t num_rays, uint32_t num_samples, const float *sigma, const float *rgb, const float *edges,
                          const tn_rgb_background *background, const float *d_out_rgb, const float *d_out_acc, float *d_sigma,
                          float *d_rgb, void *stream_) {
    return guarded([&] {
        tn::launch_composite_backward(num_rays, num_samples, sigma, rgb, edges, background_of(background), d_out_rgb, d_out_acc, d_sigma,
                                      d_rgb, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_composite(size_t num_rays, uint32_t num_samples, const float *sigma, const float *rgb, const float *edges,
                 const tn_rgb_background *background, float *out_rgb, float *out_acc, float *out_depth, float *out_weights,
                 const uint32_t *ray_index, const uint32_t *count, void *stream_) {
    return guarded([&] {
        tn::launch_composite(num_rays, num_samples, sigma, rgb, edges, background_of(background), out_rgb, out_acc, out_depth,
                             out_weights, (hipStream_t)stream_, ray_index, count);
        TN_HIP(hipGetLastError());
    });
}

int tn_gather_uint32(int elem_size, uint32_t num_values, uint32_t num_indices, const uint32_t *indices,
                     const void *values, void *result, void *stream_) {
    return guarded([&] {
        tn::launch_gather_uint32(elem_size, num_values, num_indices, indices, values, result, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

int tn_scatter_ema_uint32(int elem_size, uint32_t num_result, uint32_t num_indices, const uint32_t *indices,
                          double decay, const void *values, void *result, void *stream_) {
    return guarded([&] {
        tn::launch_scatter_ema_uint32(elem_size, num_result, num_indices, indices, decay, values, result, (hipStream_t)stream_);
        TN_HIP(hipGetLastError());
    });
}

}  // extern "C"
