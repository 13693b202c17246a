// tn_devbuf.h -- owning device buffers of libtetranerf_hip (blocking hipMalloc / hipFree: load-time structures).
#pragma once
#include "tn_common.h"

namespace tn {

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void alloc(size_t count) {
        release();
        if (count) TN_HIP(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
    }
    void upload(const std::vector<T> &h) {
        alloc(h.size());
        if (!h.empty()) TN_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
    ~DevBuf() { release(); }
};

struct DevWideBvh {
    DevBuf<float> leaf_tri, boxes;
    DevBuf<uint32_t> leaf_id, child;
    WideBvh view{};
    void set_view(size_t n_nodes, float scene_max, uint32_t leaf_w = WIDE) {
        view.leaf_w = leaf_w; view.leaf_shift = leaf_w == 16 ? 4u : (leaf_w == 32 ? 5u : 6u);
        view.leaf_tri = leaf_tri.p; view.leaf_id = leaf_id.p; view.boxes = boxes.p; view.child = child.p;
        view.n_nodes = (uint32_t)n_nodes;
        view.scene_max = scene_max;
    }
    void upload(const HostWideBvh &h, float scene_max) {
        leaf_tri.upload(h.leaf_tri);
        leaf_id.upload(h.leaf_id);
        boxes.upload(h.boxes);
        child.upload(h.child);
        set_view(h.child.size() / WIDE, scene_max, h.leaf_w);
    }
};

}  // namespace tn
