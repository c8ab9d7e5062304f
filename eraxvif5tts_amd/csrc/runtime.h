// runtime.h -- host-side plumbing shared by model.hip / vocoder.hip / ops.hip: error state, device buffers, uploads.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// owns hipMalloc'd memory; freed in the destructor of the handle that holds it
struct DevArena {
    std::vector<void*> ptrs;
    size_t total = 0;
    int alloc(void** out, size_t bytes, bool zero = true);
    template <typename T> int alloc_t(T** out, size_t count, bool zero = true) { return alloc((void**)out, count * sizeof(T), zero); }
    void release();
    ~DevArena() { release(); }
};

int f5_check_device();                       // F5_ENODEVICE unless a gfx950 device is current
int f5_cu_count();                           // compute units of the current device (cached; 256 on MI355X)
uint16_t f5_f32_to_bf16_bits(float f);       // round-to-nearest-even, NaN preserved
size_t f5_elem_size(int precision);          // 2 (bf16) / 4 (fp32)
// upload host fp32 -> device in the activation dtype of `precision` (bf16 RNE on the host) / as fp32
int f5_upload_t(DevArena& a, int precision, const float* host, size_t count, void** out);
int f5_upload_f32(DevArena& a, const float* host, size_t count, float** out);

struct TensorSlot {
    std::vector<int64_t> shape;
    std::vector<float> host;
    bool set = false;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : shape) n *= d;
        return n;
    }
};
typedef std::map<std::string, TensorSlot> SlotMap;
int f5_slot_set(SlotMap& slots, const char* name, const float* host, const int64_t* shape, int ndim);
int f5_slots_all_set(const SlotMap& slots);
