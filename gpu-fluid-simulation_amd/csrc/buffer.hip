// buffer.hip — fs_buffer: the HIP device-buffer counterpart of the reference's
// ResizableBuffer<T> / SSBO<T> (src/buffer.rs:9-173).  Same semantics: typed by
// element size, grow-only resize that preserves contents, writes trimmed to the
// buffer (logged, never failing).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <new>
#include <string>

#include "../../include/fluidsim.h"

struct fs_buffer {
    void* dev = nullptr;
    size_t elem = 0;
    size_t len = 0;
    int device = 0;
    std::string name;
};

namespace {
thread_local std::string g_buf_err;
size_t device_max_elems(size_t elem) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return (size_t)-1 / elem;
    return total_b / elem;   // counterpart of device.limits().max_buffer_size (src/buffer.rs:49)
}
}  // namespace

extern "C" {

fs_status fs_buffer_create(int device, size_t elem_size, size_t len, const char* name, fs_buffer** out) {
    if (!out || elem_size == 0) return FS_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FS_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FS_ERR_DEVICE;
    fs_buffer* b = new (std::nothrow) fs_buffer();
    if (!b) return FS_ERR_OOM;
    b->elem = elem_size;
    b->len = len;
    b->device = device;
    b->name = name ? name : "buffer";
    if (len) {
        if (hipMalloc(&b->dev, len * elem_size) != hipSuccess) { delete b; return FS_ERR_OOM; }
        if (hipMemset(b->dev, 0, len * elem_size) != hipSuccess) { (void)hipFree(b->dev); delete b; return FS_ERR_DEVICE; }
    }
    *out = b;
    return FS_OK;
}

fs_status fs_buffer_resize(fs_buffer* b, size_t new_cap, int* resized) {
    if (!b) return FS_ERR_INVALID;
    if (resized) *resized = 0;
    if (new_cap < b->len) return FS_OK;                       // src/buffer.rs:47 (returns false)
    if (hipSetDevice(b->device) != hipSuccess) return FS_ERR_DEVICE;
    const size_t max_cap = device_max_elems(b->elem);
    if (max_cap < new_cap) {                                  // src/buffer.rs:50-55
        std::fprintf(stderr,
                     "fluidsim: buffer '%s': %zu elements is the most this device can hold, %zu were asked for; "
                     "clamping the resize\n",
                     b->name.c_str(), max_cap, new_cap);
        new_cap = max_cap;
    }
    void* nd = nullptr;
    if (new_cap) {
        if (hipMalloc(&nd, new_cap * b->elem) != hipSuccess) return FS_ERR_OOM;
        if (hipMemset(nd, 0, new_cap * b->elem) != hipSuccess) { (void)hipFree(nd); return FS_ERR_DEVICE; }
        const size_t keep = b->len < new_cap ? b->len : new_cap;
        if (keep && hipMemcpy(nd, b->dev, keep * b->elem, hipMemcpyDeviceToDevice) != hipSuccess) {   // :59-63
            (void)hipFree(nd);
            return FS_ERR_DEVICE;
        }
    }
    if (b->dev) (void)hipFree(b->dev);
    b->dev = nd;
    b->len = new_cap;
    if (resized) *resized = 1;
    return FS_OK;
}

fs_status fs_buffer_write(fs_buffer* b, size_t offset, const void* data, size_t count) {
    if (!b || (!data && count)) return FS_ERR_INVALID;
    if (offset >= b->len) {
        if (count) std::fprintf(stderr, "fluidsim: write at offset %zu lies past the buffer's %zu elements; nothing written\n", offset, b->len);
        return FS_OK;
    }
    if (count > b->len - offset) {                            // src/buffer.rs:71-75, offset-aware
        std::fprintf(stderr, "fluidsim: %zu elements do not fit behind offset %zu of a %zu-element buffer; writing the part that fits\n",
                     count, offset, b->len);
        count = b->len - offset;
    }
    if (!count) return FS_OK;
    if (hipSetDevice(b->device) != hipSuccess) return FS_ERR_DEVICE;
    if (hipMemcpy((char*)b->dev + offset * b->elem, data, count * b->elem, hipMemcpyHostToDevice) != hipSuccess)
        return FS_ERR_DEVICE;
    return FS_OK;
}

fs_status fs_buffer_read(fs_buffer* b, size_t offset, void* dst, size_t count) {
    if (!b || (!dst && count)) return FS_ERR_INVALID;
    if (offset >= b->len) return FS_OK;
    if (count > b->len - offset) count = b->len - offset;
    if (!count) return FS_OK;
    if (hipSetDevice(b->device) != hipSuccess) return FS_ERR_DEVICE;
    if (hipMemcpy(dst, (const char*)b->dev + offset * b->elem, count * b->elem, hipMemcpyDeviceToHost) != hipSuccess)
        return FS_ERR_DEVICE;
    return FS_OK;
}

size_t fs_buffer_len(const fs_buffer* b) { return b ? b->len : 0; }
void* fs_buffer_device_ptr(const fs_buffer* b) { return b ? b->dev : nullptr; }

void fs_buffer_destroy(fs_buffer* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->dev) (void)hipFree(b->dev);
    delete b;
}

}  // extern "C"
