// fs_kernels.h — host-side launchers of the HIP kernels (kernels_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "fs_device.h"

namespace fsd {

void launch_reorder(hipStream_t st, const StepParams& P, const u64* pairs, const float2* pos_in, const float2* vel_in,
                    float2* pos_s, float2* vel_s, float2* pred_s, uint32_t* key_s, uint32_t* cs, uint32_t* start_ref,
                    void* work, uint32_t* counter, uint32_t work_cap, unsigned long long* safe /* kin_safe: one bit per sorted particle, a word per wave */,
                    uint32_t* force_defer /* two words per 256-particle block */, uint32_t* force_work_count /* [2] */,
                    bool cs_ready = false);
// the chunked sweep of k_force reads up to 35 candidates past a row range when it scans global memory
#define FS_PRED_SLACK 64
void launch_density(hipStream_t st, const StepParams& P, const float2* pred, const uint32_t* cs,
                    const uint32_t* start_ref, const u64* pairs, const unsigned long long* safe, float* rho,
                    float2* rho2 /* {rho, +-RN(1/rho)}: the sign is the particle's safe-operand classification */,
                    uint32_t* force_defer, uint32_t* force_work, uint32_t* force_count /* force pass: pre-registered waves */,
                    uint32_t edge_grid = 0 /* != 0: edge-first slab step, column-major ids: only the blocks of the edge columns (+1), walked by this many workgroups */);
void launch_force(hipStream_t st, const StepParams& P, const float2* pos_s, const float2* vel_s, const float2* pred,
                  const float2* rho2, const uint32_t* cs, const uint32_t* start_ref, const u64* pairs, const float2* tex,
                  float2* pos_out, float2* vel_out, const float* rho_arr, uint32_t* defer_bits /* per block */,
                  uint32_t* worklist /* per block */, uint32_t* work_count,
                  void* aos_out = nullptr /* 32-B ParticleInstance records, or none */,
                  hipStream_t side = nullptr /* second stream: the pre-registered general work runs beside the lean kernel */,
                  hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr,
                  uint32_t general_grid = 0 /* workgroups of the general kernel; 0: the full grid */,
                  uint32_t* general_hint = nullptr /* host-visible word: entries the general kernel found in its lists */,
                  uint32_t edge_grid = 0 /* != 0: the lean kernel walks only the blocks of the edge columns with this many workgroups */,
                  hipEvent_t done = nullptr /* completes with the LAST launch of the pass (its own completion signal: no marker packet) */,
                  uint32_t quad_entries = 0 /* != 0: the pre-registered list is expected to hold about this many blocks, few enough for
                                               k_force_quad (four lanes per particle) */);
// pairs != nullptr: the keys are the high words of the sorted pairs (the state of the last step; launch_reorder with
// key_s == nullptr does not store them a second time), else `key` (an uploaded state).
void launch_export_aos(hipStream_t st, uint32_t n, const float2* pos, const float2* pred, const float2* vel,
                       const float* rho, const uint32_t* key, void* out, const u64* pairs = nullptr,
                       const float2* rho2 = nullptr /* != nullptr: densities are rho2[i].x (launch_density with rho == nullptr) */);
// key[i] = pairs[i] >> 32 (pairs != nullptr), rho[i] = rho2[i].x (rho2 != nullptr): the copies the step no longer writes
void launch_keys_from_pairs(hipStream_t st, uint32_t n, const u64* pairs, uint32_t* key, const float2* rho2, float* rho);
void launch_import_aos(hipStream_t st, uint32_t n, const void* in, float2* pos, float2* pred, float2* vel, float* rho,
                       uint32_t* key);
void launch_render_density(hipStream_t st, const StepParams& P, float2 wmin, float2 wmax, uint32_t width,
                           uint32_t height, const float2* pred, const float2* vel, const uint32_t* cs,
                           const uint32_t* start_ref, const u64* pairs, float4* out);
// Obstacle push-out field (kernels_field.hip); h <= 1024, w < 65536.
void launch_gradient_field(hipStream_t st, const unsigned char* image, uint32_t w, uint32_t h, float* dist,
                           uint32_t* nearest, float2* field);
// Exhaustive proof of fs_device.h div_const_fast over lo <= |x| <= hi (both signs).
#define FS_CONSTDIV_MIN 8.67361737988403547e-19f   /* 2^-60 */
void launch_verify_constdiv(hipStream_t st, float c, float y, float lo, float hi, uint32_t* mismatches);
void launch_verify_unary(hipStream_t st, int which /* 0 rcp_rn_fast, 1 sqrt_rn_fast */, float lo, float hi,
                         uint32_t* mismatches);
size_t gap_entry_size();
void launch_fill_gaps(hipStream_t st, uint32_t* cs, const void* work, const uint32_t* counter, uint32_t work_cap);

// ---- slab (multi-GPU) mode, kernels_slab.hip -------------------------------------------------
// `out`: the counting sort's (key, ticket) words (counting_sort_kt) or, in bitonic mode, the pairs; `blockcnt`: one uint2 per
// 256-slot block; `stage`: slab_stage_words(capacity) words; `state`: one u64 per slab_msg_groups(capacity) (look-back);
// `epoch`: a number unique to this launch among the handle's launches (the tick).  counters[6] is the look-back's ticket.
void launch_slab_pack(hipStream_t st, const StepParams& P, uint32_t main_slots, uint32_t R, int has_left,
                      int has_right, const float2* pos, const float2* vel, const unsigned char* owned, u64* out,
                      uint32_t* hist, void* blockcnt, uint32_t* stage, void* state, uint32_t epoch,
                      void* msg_left, void* msg_right, uint32_t* counters, uint32_t* gap_counter, unsigned long long* safe,
                      bool counting, bool overlap = false /* the slots past main_slots hold last step's migrants (strip step) */,
                      bool lists = true /* false: the messages were pre-built (launch_slab_prepack): only check that every particle
                                           the full classification flags was in the edge zone [own_lo, prev_adv_lo) u [prev_adv_hi, own_hi)
                                           of the last step (key_prev: its sorted keys) */,
                      const uint32_t* key_prev = nullptr, uint32_t prev_adv_lo = 0, uint32_t prev_adv_hi = 0,
                      bool skip_edge = false /* with lists == false: the slots of the last step's edge-zone particles were classified by
                                                launch_slab_prepack(classify) already — leave them alone */);
// Edge-first step: the NEXT step's messages from the particles the StepParams::adv_outside force launch has just advanced.
// `P_next`: window + tick constants of the next pack, adv_* as in that force launch; `epoch` unique among the handle's k_slab_msg launches.
void launch_slab_prepack(hipStream_t st, const StepParams& P_next, uint32_t cap, uint32_t R, int has_left, int has_right,
                         const float2* pos, const float2* vel, const unsigned char* owned, const uint32_t* key_s, void* blockcnt,
                         uint32_t* stage, void* state, uint32_t epoch, void* msg_left, void* msg_right, uint32_t* counters,
                         const uint32_t* cs, uint32_t edge_grid = 0 /* != 0 (column-major ids): walk only the edge columns' blocks */,
                         bool classify = false /* also do the next launch_slab_pack's work for the slots of the particles it takes:
                                                  key, histogram ticket (`counting`) and out[] entry, the lost counter */,
                         bool counting = false, uint32_t main_slots = 0, u64* out = nullptr, uint32_t* hist = nullptr);
// Overlapped slab step — the boundary strips (kernels_slab.hip).  `P` = the main array's StepParams; win[4] = the two strip
// windows as LOCAL column ranges [win[0], win[1]) and [win[2], win[3]); strip_counters: [0] live strip particles (written by the
// strip's scan), [1] slots filled from the main array, [2] slots in use.
void launch_strip_gather(hipStream_t st, const StepParams& P, const uint32_t win[4], uint32_t R, uint32_t strip_cap,
                         const uint32_t* cs, uint32_t* rowbase /* 2 * grid_h */, const u64* pairs, const float2* pos_s,
                         const float2* vel_s, float2* sp_pos, float2* sp_vel, u64* kt, uint32_t* hist, uint32_t* back,
                         unsigned long long* safe, uint32_t* strip_counters, uint32_t* counters);
void launch_strip_unpack(hipStream_t st, const StepParams& P, uint32_t main_slots, uint32_t R, uint32_t strip_cap,
                         const void* msg_left, const void* msg_right, float2* sp_pos, float2* sp_vel, u64* kt, uint32_t* hist,
                         uint32_t* back, const uint32_t* strip_counters, uint32_t* counters);
void launch_strip_writeback(hipStream_t st, const StepParams& P_strip, uint32_t main_slots, uint32_t strip_cap, const u64* sp_pairs,
                            const uint32_t* back, const float2* sp_pos_out, const float2* sp_vel_out, const float2* sp_pred,
                            const float* sp_rho, float2* pos, float2* vel, float2* pred, float* rho, uint32_t* key,
                            unsigned char* owned, uint32_t* counters);
size_t slab_stage_words(uint32_t cap);
size_t slab_msg_groups(uint32_t cap);
void launch_slab_unpack(hipStream_t st, const StepParams& P, uint32_t main_slots, uint32_t R, const void* msg_left,
                        const void* msg_right, float2* pos, float2* vel, u64* out, uint32_t* hist, uint32_t* counters,
                        bool counting);
void launch_slab_reorder(hipStream_t st, const StepParams& P, uint32_t cap, const u64* pairs, const float2* pos_in,
                         const float2* vel_in, float2* pos_s, float2* vel_s, float2* pred_s, uint32_t* key_s,
                         unsigned char* owned, uint32_t* cs, uint32_t* start_ref, void* work, uint32_t* counter,
                         uint32_t work_cap, uint32_t* n_live_out, unsigned long long* safe, uint32_t* force_defer,
                         uint32_t* force_work_count);
void launch_slab_export(hipStream_t st, const StepParams& P, uint32_t cap, const float2* pos, const float2* pred,
                        const float2* vel, const float* rho, const uint32_t* key, void* out);
void launch_slab_import(hipStream_t st, const StepParams& P, uint32_t n, uint32_t cap, const void* in, float2* pos,
                        float2* pred, float2* vel, float* rho, uint32_t* key, unsigned char* owned);
// migr_first / migr_count: the migrant slots of an overlapped step (outside the sorted prefix), or 0 / 0
void launch_slab_colhist(hipStream_t st, const StepParams& P, const uint32_t* cs, uint32_t* hist_global, uint32_t migr_first = 0,
                         uint32_t migr_count = 0, const unsigned char* owned = nullptr, const uint32_t* key = nullptr);
void launch_slab_maxspeed(hipStream_t st, const uint32_t* n_live, const float2* vel, const unsigned char* owned,
                          uint32_t* out_bits, uint32_t migr_first = 0, uint32_t migr_count = 0);
size_t slab_message_bytes(uint32_t R);

// Bitonic network of sort.wgsl:27-51 / simulation.rs:323-347 on (key<<32 | index) pairs.
// Returns the number of kernel launches issued.
// `dirty`: one u32 per 4096-element tile (sort_tile_count(n) entries), scratch owned by the caller.
// keygen != nullptr: the init pass computes the pairs from pos/vel itself (predict + key fused in).
// Late-stage plan of one sort call (kernels_sort.hip, k_late_cert).  Whatever the plan, the result is the network's.
struct SortPlan {
    int fuse_stage = -1;           // < 0: default stage, 0: per-stage launches only, k: the shifted merge from stage k
    int fallback = 0;              // what stands by for a failing certificate: 0 the per-stage launches (each returns at
                                   // once when not needed, ~5 us apiece), 1 one persistent launch (slow when it has work)
    uint32_t* feedback = nullptr;  // host-visible words the certificate reports to (stage, verdict, fit class, seq), or none
    uint32_t seq = 0;
    int inject_timeout = 0;        // tests (FS_SORT_INJECT_TIMEOUT=1): the stand-by kernel reports a barrier time-out it did not have
};
// keygen3d != nullptr (3D engine): the same fusion with float4 pos / vel and the 3D cell key.
struct KeyGen3 { float dt, h, bx, by, bz; uint32_t gw, gh; };
int launch_bitonic_sort(hipStream_t st, u64* pairs, uint32_t n, uint32_t* dirty, const StepParams* keygen = nullptr,
                        const float2* pos = nullptr, const float2* vel = nullptr, uint32_t* gap_counter = nullptr,
                        const SortPlan* plan = nullptr, const KeyGen3* keygen3d = nullptr, const float4* pos4 = nullptr,
                        const float4* vel4 = nullptr);
// dirty[sort_plan_word(n) ..]: [0] verdict of the last certificate, [1] / [2] shifted-merge / per-stage plan counters,
// [3] fallback barrier, [4] fallback barrier time-outs, [5] fit class
uint32_t sort_plan_word(uint32_t n);
#define FS_SORT_NO_PLAN 255u
uint32_t sort_tile_count(uint32_t n);

// FS_SORT_COUNTING (kernels_csort.hip).  The scratch must be all-zero when the handle is created (histogram, tickets);
// every step leaves it that way.  `epoch`: a number unique to the launch among the handle's launches.
//   single domain: launch_counting_sort (hist -> scan -> scatter: fills `cs`) + launch_counting_reorder (rank fix-up fused
//   with the whole reorder pass: fills `pairs`, pos_s / vel_s / pred_s, start_ref, safe bits);
//   slabs: k_slab_pack / k_slab_unpack fill the histogram, then launch_counting_sort_pairs (scan -> scatter; live count)
//   + launch_counting_reorder_slab.
size_t counting_sort_scratch_words(uint32_t n, uint32_t ncell_max);
u64* counting_sort_kt(uint32_t* scratch, uint32_t n, uint32_t ncell_alloc);
uint32_t* counting_sort_hist(uint32_t* scratch);
void launch_counting_sort(hipStream_t st, const StepParams& P, const float2* pos, const float2* vel, uint32_t* cs,
                          uint32_t* scratch, uint32_t* gap_counter, unsigned long long* safe, uint32_t epoch);
void launch_counting_reorder(hipStream_t st, const StepParams& P, uint32_t* scratch, u64* pairs, const uint32_t* cs,
                             const float2* pos_in, const float2* vel_in, float2* pos_s, float2* vel_s, float2* pred_s,
                             uint32_t* key_s, uint32_t* start_ref, unsigned long long* safe, uint32_t* force_defer,
                             uint32_t* force_work_count);
// n_dev (may be null): device word holding the number of slots in use (<= cap); the grids still cover `cap`
void launch_counting_sort_pairs(hipStream_t st, uint32_t cap, uint32_t ncell, uint32_t ncell_alloc, uint32_t* cs, uint32_t* scratch,
                                uint32_t* n_live_out, uint32_t epoch, const uint32_t* n_dev = nullptr,
                                unsigned long long* safe_preset = nullptr /* the slots' "safe operand" words, set to all-ones for k_cs_fixreorder */);
void launch_counting_reorder_slab(hipStream_t st, const StepParams& P, uint32_t cap, uint32_t ncell_alloc, uint32_t* scratch, u64* pairs,
                                  const uint32_t* cs, const float2* pos_in, const float2* vel_in, float2* pos_s, float2* vel_s,
                                  float2* pred_s, uint32_t* key_s, unsigned char* owned, uint32_t* start_ref,
                                  unsigned long long* safe, uint32_t* force_defer, uint32_t* force_work_count,
                                  const uint32_t* n_dev = nullptr, hipEvent_t done = nullptr /* signalled by the kernel's completion */);

}  // namespace fsd
