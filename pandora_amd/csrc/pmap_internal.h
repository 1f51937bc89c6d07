// Internal definitions shared by the host and device halves of libpandora_amd.
#ifndef PMAP_INTERNAL_H
#define PMAP_INTERNAL_H

#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "pandora_amd.h"

// One CSR record (a MiniRecord, include/minirecord.h:9-14, with the path replaced by its rank).
struct PmapRec {
    uint32_t prg;
    uint32_t knode_strand;  // knode_id << 1 | strand
    uint32_t rank;          // rank of (prg, path) in (prg, prg::Path::operator<) order
    uint32_t orig;          // record index in the caller's arrays
};

// One open-addressing slot of the minimizer table: 32 bytes, one probe = one slot.  The key's first
// record rides in the slot: random 16-byte fetches beyond the caches are the scarce resource on a big
// index (~35 G/s on one MI355X), and nearly every key has a single record, so the hit emission
// hardly ever has to fetch from the record array.
struct PmapSlot {
    uint64_t key;   // ~0 = empty
    uint32_t off;   // first record in the CSR array
    uint32_t cnt;   // number of records
    PmapRec first;  // copy of recs[off]
};
static_assert(sizeof(PmapSlot) == 32, "a slot is two 16-byte loads inside one 64-byte sector");

static const uint64_t PMAP_EMPTY_KEY = ~0ULL;

// Home slot of a key.  Keys are hash64 outputs already, one 32-bit multiply spreads them enough.
#ifdef __HIPCC__
#define PMAP_HD __host__ __device__
#else
#define PMAP_HD
#endif
PMAP_HD static inline uint32_t pmap_key_mix(uint64_t key) {
    return ((uint32_t) key ^ (uint32_t) (key >> 32)) * 2654435761u;
}
PMAP_HD static inline uint32_t pmap_slot_of(uint64_t key, uint32_t log2cap) {
    return pmap_key_mix(key) >> (32 - log2cap);
}
// Presence filter.  `fb` = log2 of the number of bits of the filter in its low byte (0 = no filter) plus
// flags.  Hashed form: bit pmap_key_mix(key) >> (32 - bits); with PMAP_FILTER_TWO every key sets two
// bits of its 32-bit word instead of one (denser indexes: the false-positive rate is about the square of
// the fill instead of the fill).  PMAP_FILTER_DIRECT: bit number = the key itself, for keys below 2^bits
// (k <= 15: 2^30 bits = 128 MiB) - exact, so the table is read only for keys it holds, and since
// minimizers are window minima their keys crowd the low end of the range, which keeps the busy part of
// the bitmap in the caches however many keys the index has.
static const uint32_t PMAP_FILTER_TWO = 0x100u;
static const uint32_t PMAP_FILTER_DIRECT = 0x200u;
PMAP_HD static inline uint32_t pmap_filter_bit(uint64_t key, uint32_t hm, uint32_t fb) {
    return (fb & PMAP_FILTER_DIRECT) ? (uint32_t) key : hm >> (32 - (fb & 0xFFu));
}
PMAP_HD static inline uint32_t pmap_filter_word(uint64_t key, uint32_t hm, uint32_t fb) {
    return pmap_filter_bit(key, hm, fb) >> 5;
}
PMAP_HD static inline uint32_t pmap_filter_mask(uint64_t key, uint32_t hm, uint32_t fb) {
    uint32_t m = 1u << (pmap_filter_bit(key, hm, fb) & 31u);
    if (fb & PMAP_FILTER_TWO) m |= 1u << ((hm >> 3) & 31u);      // hash bits the word index does not use
    return m;
}

// A number no other index object of this process ever gets: caches keyed on an index compare this, not
// the object's address (the allocator hands the address of a destroyed index to the next one).
inline uint64_t pmap_new_serial() {
    static std::atomic<uint64_t> next{0};
    return ++next;
}

struct pmap_index {
    const uint64_t serial = pmap_new_serial();
    uint32_t k = 0;
    uint32_t n_prg = 0;
    uint64_t n_keys = 0, n_recs = 0, n_paths = 0;
    uint32_t log2cap = 0;
    uint32_t max_cnt = 0;       // most records under one key
    // host copies, caller's record numbering
    std::vector<uint64_t> key_of_rec;
    std::vector<uint32_t> rec_prg, rec_knode, path_rank;
    std::vector<uint8_t> rec_strand;
    std::vector<uint64_t> path_off;
    std::vector<uint32_t> iv_start, iv_len;
    std::vector<uint32_t> min_path_length;
    // host copies of the device structures (kept for checks and rebuilds)
    std::vector<PmapSlot> slots;
    std::vector<PmapRec> recs;
    // presence filter: pmap_filter_mask(key) is set in word pmap_filter_word(key) for every key; 0 = no filter
    uint32_t filter_bits = 0;
    std::vector<uint32_t> filter;
    // device copies, one per GPU the index has been used on (uploaded on first use, under `upload_mu`)
    struct OnDevice {
        PmapSlot *slots = nullptr;
        PmapRec *recs = nullptr;
        uint32_t *filter = nullptr;
    };
    static const int MAX_DEVICES = 64;
    OnDevice dev[MAX_DEVICES];
    std::mutex upload_mu;
    uint64_t device_bytes = 0;      // per device
};

// pinned host memory (hipHostMalloc), nullptr on failure
int pmap_result_hit_ranks(const pmap_result *res, uint32_t *rank);
void *pmap_pinned_alloc(size_t bytes);
void pmap_pinned_free(void *p);

void pmap_set_error(const std::string &msg);
int pmap_fail(int code, const std::string &msg);

// host side of index construction (pmap_index.cpp); uploads via pmap_index_upload (pmap_device.hip)
int pmap_index_build_host(const pmap_index_desc_t *desc, pmap_index *idx);
int pmap_index_upload(pmap_index *idx);          // to the calling thread's current device
void pmap_index_release_device(pmap_index *idx);

// The mapping contexts (opaque) pmap_pangraph_from_read_file spreads its batches over: those of
// pmap_init_devices, or three on the calling thread's GPU.  pmap_bind_context makes one of them the calling
// thread's current context.
int pmap_contexts_for_file(std::vector<void *> *out);
int pmap_bind_context(void *ctx);

// (float) threshold of one prg, src/utils.cpp:211-213,219-220
static inline uint32_t pmap_threshold(uint32_t mpl, uint32_t expected, float fraction, uint32_t min_cluster_size) {
    uint32_t m = mpl < expected ? mpl : expected;
    uint32_t length_based = (uint32_t) ((float) m * fraction);
    return length_based > min_cluster_size ? length_based : min_cluster_size;
}

#endif
