// snn_h5.hip -- host-only helper of the database reader (supernnova_amd/utils/h5reader.py): gathers the
// elements of a variable-length HDF5 dataset -- the light curves of database.h5, written by utils/data_utils.py:691-733
// as `hf["data"][idx] = np.ravel(arr)` -- from the file's global heap into ONE caller buffer (pinned host memory for
// the GPU feed), instead of one Python object per light curve.  Format per the HDF5 File Format Specification 3.0:
// a dataset element is a 16-byte record {uint32 length (elements), uint64 collection address, uint32 object index};
// a collection is "GCOL", version, 3 reserved bytes, uint64 size, then objects {uint16 index, uint16 refcount,
// uint32 reserved, uint64 size, data padded to 8 bytes}, index 0 = free space.
#include <stdint.h>
#include <string.h>
#include <unordered_map>
#include <vector>

#include "snn_host.h"

namespace {
struct HeapObj { int64_t off, size; };
struct Collection { std::vector<HeapObj> by_index; };

inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
}

extern "C" int snn_h5_vlen_gather(const void* file_v, int64_t file_size, const void* desc_v, int64_t n, int32_t elem_size,
                                  void* out_v, int64_t out_bytes)
{
    const uint8_t* file = (const uint8_t*)file_v;
    const uint8_t* desc = (const uint8_t*)desc_v;
    uint8_t* out = (uint8_t*)out_v;
    if (!file || !desc || (!out && out_bytes > 0) || n < 0 || elem_size < 1 || file_size < 16) {
        snn_set_error("snn_h5_vlen_gather: null pointer or bad size");
        return SNN_ERR_ARG;
    }
    std::unordered_map<uint64_t, Collection> cols;
    int64_t at = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* d = desc + 16 * i;
        const int64_t len = rd32(d);
        const uint64_t addr = rd64(d + 4);
        const uint32_t idx = rd32(d + 12);
        // an empty element, or one whose heap address is 0 / "undefined" (all ones), holds nothing -- as h5reader._heap_object
        if (len == 0 || addr == 0 || addr == ~(uint64_t)0) continue;
        const int64_t bytes = len * elem_size;
        if (bytes > out_bytes - at) { snn_set_error("snn_h5_vlen_gather: output buffer too small"); return SNN_ERR_ARG; }
        auto it = cols.find(addr);
        if (it == cols.end()) {
            // every bound is tested by subtraction from a quantity already known to be in range: the fields come from the
            // file and a sum such as addr + 16 wraps for addresses near 2^64
            const uint64_t fsz = (uint64_t)file_size;
            if (addr > fsz - 16 || memcmp(file + addr, "GCOL", 4) != 0) {
                snn_set_error("snn_h5_vlen_gather: element %lld points to %llu, which is not a global heap collection",
                              (long long)i, (unsigned long long)addr);
                return SNN_ERR_ARG;
            }
            const uint64_t csize = rd64(file + addr + 8);
            if (csize < 16 || csize > fsz - addr) { snn_set_error("snn_h5_vlen_gather: collection runs past the end of the file"); return SNN_ERR_ARG; }
            Collection c;
            const uint64_t end = addr + csize;      // <= file_size: no wrap
            uint64_t p = addr + 16;
            while (end - p >= 16) {                 // p <= end throughout
                const uint16_t oi = rd16(file + p);
                const uint64_t osz = rd64(file + p + 8);
                if (oi == 0) break;                 // free space: by the format the last object of a collection
                if (osz > end - p - 16) { snn_set_error("snn_h5_vlen_gather: heap object runs past its collection"); return SNN_ERR_ARG; }
                if (c.by_index.size() <= oi) c.by_index.resize(oi + 1, HeapObj{-1, 0});
                c.by_index[oi] = HeapObj{(int64_t)(p + 16), (int64_t)osz};
                const uint64_t adv = 16 + ((osz + 7) & ~(uint64_t)7);     // osz <= end - p - 16 < 2^63: no wrap
                if (adv > end - p) break;           // padding of the last object may run to the collection's end
                p += adv;
            }
            it = cols.emplace(addr, std::move(c)).first;
        }
        const Collection& c = it->second;
        if (idx >= c.by_index.size() || c.by_index[idx].off < 0 || c.by_index[idx].size < bytes) {
            snn_set_error("snn_h5_vlen_gather: element %lld: heap object %u missing or shorter than %lld bytes", (long long)i, idx,
                          (long long)bytes);
            return SNN_ERR_ARG;
        }
        memcpy(out + at, file + c.by_index[idx].off, (size_t)bytes);
        at += bytes;
    }
    return SNN_OK;
}
