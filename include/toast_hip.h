/* toast_hip.h -- C ABI of the MI355X-native TOAST map-making hot path (libtoast_hip.so).
 *
 * This is the drop-in boundary: every entry point replaces one binding of the reference's
 * pybind11 extension `toast._libtoast` (or one method of its OpenMP-target memory manager)
 * and is cited below as  [ref: <file>:<line>]  with paths relative to
 * /root/reference/src/toast/_libtoast/.  Plain pointers and sizes only; no C++ or torch
 * types.  INTEGRATION.md shows the reference-side binding a TOAST maintainer would add.
 *
 * Two levels:
 *
 *   toast_hip_<kernel>(..., int use_accel)
 *       Same arguments, same meaning and same validation rules as the reference binding:
 *       all pointers are HOST pointers.  With use_accel != 0 the large arrays must have been
 *       registered with toast_hip_accel_create() + toast_hip_accel_update_device() (exactly
 *       like OmpManager) and the kernel runs on their device copies, asynchronously on the
 *       library stream.  With use_accel == 0 the buffers are staged to the GPU, processed
 *       and copied back before the call returns.  There is no CPU implementation in this
 *       library: without a usable gfx950 device every call fails (return code != 0).
 *
 *   toast_hip_<kernel>_dev(..., void *stream)
 *       Large arrays are DEVICE pointers (hipMalloc'd by the caller, e.g. torch tensors);
 *       small per-call arrays (index arrays, intervals, per-detector scalars) remain host
 *       pointers and are packed into one cached parameter block.  Asynchronous on `stream`
 *       (a hipStream_t, NULL = default stream).
 *
 *   Beyond the one-to-one replacements, the _dev level carries the entry points of the solver
 *   loop that have no single counterpart in `toast._libtoast` (each cites the reference code it
 *   stands in for): the fused offset-template left-hand side (toast_hip_offset_accumulate_dev /
 *   _scan_project_dev), pointing evaluated inside the kernels for the reference's uncached
 *   full_pointing=False mode (toast_hip_otf_* with the toast_hip_otf_pointing descriptor, the
 *   compact pixel cache, the HWP table), the amplitude-vector algebra of the PCG
 *   (toast_hip_vec_*), solver flags and Offset-template set-up
 *   (toast_hip_combine_flags_dev, toast_hip_offset_count_flagged_dev).
 *
 * "Optional" arrays follow the reference convention: an array whose length differs from
 * n_samp is treated as absent (ops_pixels_healpix.cpp:1204-1211,
 * ops_mapmaker_utils.cpp:181-197).
 *
 * Error handling: every function returns 0 on success and a nonzero code on failure;
 * toast_hip_last_error() returns the message (thread local).  The reference throws
 * std::runtime_error with the same text where one exists.
 */
#ifndef TOAST_HIP_H
#define TOAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOAST_HIP_OK 0
#define TOAST_HIP_ERR_ARG 1      /* invalid argument / shape */
#define TOAST_HIP_ERR_DEVICE 2   /* no device, HIP runtime error */
#define TOAST_HIP_ERR_MEMORY 3   /* memory manager: not present / already present / size */

/* Sample interval, 32 bytes.  The kernels process first <= isamp < last.
 * [ref: intervals.hpp:10-15; dtype intervals.cpp:11-38] */
typedef struct toast_hip_interval {
    double start;
    double stop;
    int64_t first;
    int64_t last;
} toast_hip_interval;

const char * toast_hip_last_error(void);
const char * toast_hip_version(void);

/* ------------------------------------------------------------------------------------
 * Device selection and the host-pointer -> device-pointer memory manager
 * [ref: accelerator.hpp:73-159 (class OmpManager), accelerator.cpp:233-766, Python
 *  bindings accelerator.cpp:768-1110]
 * ---------------------------------------------------------------------------------- */

/* Number of usable HIP devices > 0.  [ref: accelerator.cpp:770-779 accel_enabled] */
int toast_hip_accel_enabled(void);
/* Counter bumped by every create / adopt / delete / assign_device of the memory manager: while it
 * is unchanged, device pointers obtained earlier are still the ones the manager would return
 * (lets a caller replay a prepared launch sequence without looking everything up again). */
int toast_hip_accel_generation(uint64_t * generation);
/* The device memory arena [ref: OmpPoolResource, accelerator.cpp:13-230; the pool of `mem_gb` taken at assign_device,
 * accelerator.cpp:292-303 (dormant upstream)].  Every device block of the library -- registered arrays, temporaries,
 * the solver's packed pointing cache, FFT work space -- is a range of a slab that was taken from the driver with one
 * hipMalloc and touched once; released blocks go back to the arena, not to the driver, so that after set-up no operator
 * calls hipMalloc or hipFree.  TOAST_HIP_ALLOC=plain restores one hipMalloc per block; TOAST_HIP_ARENA_SLAB_GB (16) is
 * the size of a slab taken on demand; TOAST_HIP_ARENA_RESERVE_GB overrides assign_device's mem_gb. */
typedef struct toast_hip_arena_stats_t {
    int64_t slabs;            /* slabs held now */
    uint64_t slab_bytes;      /* ... their bytes */
    uint64_t used_bytes;      /* bytes in live blocks */
    uint64_t peak_used_bytes;
    int64_t slab_mallocs;     /* hipMalloc calls for slabs so far */
    int64_t slab_frees;       /* slabs given back to the driver */
    double malloc_ms;         /* wall time inside hipMalloc (slabs, and direct blocks with TOAST_HIP_ALLOC=plain) */
    double max_malloc_ms;     /* ... the longest single call */
    double touch_ms;          /* host time enqueueing the first-touch fills of new slabs */
    int64_t allocs;           /* blocks handed out */
    int64_t releases;
    int64_t direct_mallocs;   /* blocks that went straight to hipMalloc */
    int64_t failed;           /* requests that could not be served */
    /* slabs of 8 GB and more are built from 1 GB physical chunks of two different HBM zones mapped alternately
     * (csrc/vmm_slab.cpp; TOAST_HIP_ARENA_INTERLEAVE=0 turns that off): */
    int64_t interleaved_slabs;  /* such slabs alive */
    int64_t chunks;             /* chunks mapped into them so far */
    int64_t chunks_other_zone;  /* ... of which from a zone other than the slab's first chunk (ideal: half) */
    int64_t chunks_created;     /* chunks created while searching for the second zone (the surplus is released) */
    double interleave_ms;       /* wall time building them (part of malloc_ms) */
    double same_zone_tbs;       /* reference rate of the zone measurement (TB/s) */
    /* how the searches for chunks of the other zone ended (budget counted in measuring passes and candidate bytes:
     * TOAST_HIP_ARENA_SEARCH_PROBES, TOAST_HIP_ARENA_SEARCH_GB; TOAST_HIP_ARENA_SEARCH_MS is a hard cap only): */
    int64_t chunks_other_wanted;  /* slots that the pattern P Q Q P gives to the other zone (placement ok: chunks_other_zone equals it) */
    int64_t searches;             /* slabs whose chunks were searched for */
    int64_t searches_exhausted;   /* ... that ran out of a budget before both classes were full */
    int64_t searches_capped_ms;   /* ... of which by the hard cap in wall time */
    int64_t probes;               /* measuring passes (three kernel launches each) */
    int64_t probes_by_clock;      /* ... timed by the device's constant-rate clock inside the kernel (the rest: HIP events) */
    double create_ms_per_chunk;   /* last search: average hipMemCreate time (0.1 ms on clean memory, 20-50 while the driver clears) */
    double search_ms;             /* wall time of the searches (part of interleave_ms) */
    int64_t slabs_third_zone;     /* slabs whose BOTH chunk classes are clear of the read-mostly slab: a written timestream there
                                   * shares an HBM zone with none of the streams the sweeps read (csrc/vmm_slab.cpp) */
    int64_t read_mostly_zones;    /* HBM zones the last survey found in the range the read-mostly arrays fill (1-3; 0: none
                                   * surveyed): both chunk classes leave the readers' zone only when it is ONE zone throughout */
} toast_hip_arena_stats_t;
int toast_hip_arena_stats(toast_hip_arena_stats_t * out);
/* Did the zone placement work out?  placement_ok = 1: an interleaved slab stands and every slot meant for the other HBM
 * zone holds a chunk that measured clear of the read-mostly slabs; search_exhausted = 1: some search ran out of its budget
 * first (the slab was built from what there was: written timestreams and maps may share a zone with what the kernels
 * stream, worth 5-12 % of scan_map / build_noise_weighted).  A caller that cares can release the cached slabs
 * (toast_hip_accel_release_cached) and reserve again (toast_hip_arena_reserve_streamed) when the device is quiet.  Waits
 * for a slab that toast_hip_accel_assign_device is still building.  Any out pointer may be NULL.
 * [the pool of accelerator.cpp:292-303 has no such notion: one hipMalloc, wherever the driver puts it] */
int toast_hip_arena_placement_status(int * placement_ok, int * search_exhausted, int64_t * chunks_other_zone,
                                     int64_t * chunks_other_wanted);
/* Make the arena hold at least `bytes` (one more slab for the difference, at most 90 % of what the device has free). */
int toast_hip_arena_reserve(size_t bytes);
/* The part of the arena that serves streamed blocks and scatter targets (below): make it hold a FREE RANGE of at least
 * `bytes` (a new zone-interleaved slab of that size when it has none). */
int toast_hip_arena_reserve_streamed(size_t bytes);
/* Where [device_ptr, device_ptr + bytes) lies: interleaved = 1 when inside a zone-interleaved slab, and then how many of
 * the 1 GB chunks it touches are of the read-mostly slabs' zone / of the other zone (csrc/vmm_slab.cpp: pattern
 * P Q Q P ...).  For a slab that was built before any read-mostly array existed the two classes are relative to the
 * slab's FIRST chunk ("own" = the first chunk's class, "other" = the second class), and scatter targets go to whichever
 * of the two measured clear of the arrays when the first one was asked for -- one run of ONE class either way.
 * Waits for a slab that toast_hip_accel_assign_device is still building. */
int toast_hip_arena_block_zone(const void * device_ptr, size_t bytes, int * interleaved, int * chunks_own_zone,
                               int * chunks_other_zone);
/* The rule that sorts the candidate chunks of a zone-interleaved slab into "same zone as the reference" and "another zone",
 * applied to given rates (no device needed): the middle of the largest relative gap between neighbouring rates -- a gap counts
 * when it is at least TOAST_HIP_ARENA_ZONE_GAP (3.5 %) wide AND its upper side lies above (1 + that) x `level`, the rate a range
 * shows with itself -- or, with one level only, 0 (every rate is "another zone") / 1e300 (every rate is "the same"). */
int toast_hip_arena_zone_threshold(const double * rates, int n, double level, double * threshold);
/* The sub-allocation logic exercised on HOST memory (no device needed): n_ops random allocations / releases with the
 * bookkeeping and the contents of every live block checked after each step.  0 = sound. */
int toast_hip_arena_selftest(uint64_t seed, int n_ops, size_t granule, size_t slab_bytes, size_t max_block);
/* Raw device memory from the arena (flags = -1, also -2: a read-mostly block; -3 = a STREAMED block: a timestream that
 * sweeps read and write, placed in a slab whose 1 GB chunks come from two HBM zones in the pattern P Q Q P ... so that its
 * rows are spread over both -- such sweeps run at 6.1 instead of 5.1 TB/s there; -4 = a SCATTER TARGET: a map or amplitude
 * vector that kernels add to with atomics, placed inside one Q Q run (up to 2 GB in the zone the read-mostly blocks are NOT
 * in) -- build_noise_weighted 5.2 instead of 5.8 ms; csrc/vmm_slab.cpp, profiles/r04_a.  The slab for -3 / -4 is reserved
 * by toast_hip_accel_assign_device (TOAST_HIP_ARENA_STREAM_GB) or toast_hip_arena_reserve_streamed; blocks that do not
 * fit it are read-mostly blocks), a plain hipMalloc (0) or explicit hipExtMallocWithFlags flags (4 = physically
 * contiguous): what bench.py and the solver's packed cache use, so that they get the blocks the operators get.  Blocks
 * from flags < 0 are released with toast_hip_device_free / toast_hip_device_release. */
int toast_hip_device_malloc(size_t nbytes, int flags, void ** out);
/* Time (ms) of a read + write pass over [p, p + bytes) with the timestream kernels' access pattern (1024 rows in
 * flight): placement experiments on ranges of arena blocks. */
int toast_hip_probe_stream(void * p, size_t bytes, double * ms);
/* Experiment (csrc/vmm_slab.cpp vmm_pair_matrix): rate of a pass over two 1 GB ranges for every (physical chunk,
 * virtual slot) combination; out[n_phys * n_slots] in TB/s. */
int toast_hip_exp_vmm_pair_matrix(int n_phys, int n_slots, double * out);
/* The same pass with the 1024 rows dealt round-robin to nb <= 4 separate ranges of bytes_each. */
int toast_hip_probe_stream_split(void * const * bases, int nb, size_t bytes_each, double * ms);
/* Measurement aid (bench.py roofline.stream_ceiling): the byte mix of scan_map -- 8 B of pixels, 24 B of weights and 8 B of a
 * timestream read, 8 B written per detector-sample -- or, with d_out = NULL, of build_noise_weighted (40 B read) as plain
 * streams in the launch shape of the two kernels, no gather, no scan, no atomics: what the arrays' places in HBM let a
 * kernel of this mix reach.  [n_det, n_samp] rows, n_samp even, rows 16-byte aligned; asynchronous on `stream`. */
int toast_hip_probe_byte_mix_dev(const int64_t * d_pixels, const double * d_weights, const double * d_tod, double * d_out,
                                 int64_t n_det, int64_t n_samp, void * stream);
/* Free and total device memory of this process' GPU (hipMemGetInfo; free ranges of the arena's slabs count as free). */
int toast_hip_accel_mem_info(size_t * free_bytes, size_t * total_bytes);
/* Give the arena's slabs that hold no live block back to the driver. */
int toast_hip_accel_release_cached(void);
/* Give a block back: to the arena when it came from it (after the manager's stream has drained), else hipFree. */
int toast_hip_device_free(void * p);
int toast_hip_device_release(void * p, size_t nbytes);
/* Experiment: virtual range backed by chunk_mb-sized physical allocations mapped in order / shuffled
 * (tools/exp_alloc_flags.py, profiles/r02_d_placement_experiments.txt).  The range is never released. */
int toast_hip_device_malloc_vmm(size_t nbytes, int chunk_mb, int shuffled, void ** out);
/* Switches at run time (tools/exp_*.py, tests): key "det_major" = 0 / 1 (workgroup order of the
 * accumulate / scan kernels; the environment variable TOAST_HIP_DET_MAJOR sets the start-up value);
 * key "pair" = 0 / 1 (two detectors per workgroup in the scatter and pixel kernels: merged atomics and one
 * pixel evaluation for a co-pointing pair; start-up value from TOAST_HIP_PAIR, default 1);
 * key "vec2" = 0 / 1 (two consecutive samples per lane -- 16-byte lane accesses -- in scan_map,
 * build_noise_weighted and noise_weight when n_samp is even; start-up value from TOAST_HIP_VEC2, default 1). */
int toast_hip_set_tuning(const char * key, int value);

/* Pick this process's GPU: device = node_rank / ceil(node_procs / n_device).  `disabled`
 * != 0 refuses all later use_accel work.  Clears previously registered buffers.
 * [ref: accelerator.cpp:233-306 assign_device] */
int toast_hip_accel_assign_device(int node_procs, int node_rank, double mem_gb, int disabled);

/* Assigned device id (-1 = disabled).  Error if assign_device was never called.
 * [ref: accelerator.cpp:308-317 get_device] */
int toast_hip_accel_get_device(int * device);

/* *present = 1 if `host` is registered.  Error if registered with a different size.
 * [ref: accelerator.cpp:656-695 present] */
int toast_hip_accel_present(const void * host, size_t nbytes, int * present);

/* Allocate a device buffer of nbytes keyed by the host base pointer.  Error if the key is
 * already present.  [ref: accelerator.cpp:327-388 create] */
int toast_hip_accel_create(const void * host, size_t nbytes, const char * name);
/* The same with a statement of what the array is to the kernels, which decides where in HBM the arena puts it (see
 * toast_hip_device_malloc): kind 0 = read-mostly (what toast_hip_accel_create does), 1 = a timestream that sweeps read
 * AND write (flags = -3 there), 2 = the target of a scatter with atomics: a map, an amplitude vector (flags = -4). */
int toast_hip_accel_create_kind(const void * host, size_t nbytes, const char * name, int kind);

/* Zero the device copy.  [ref: accelerator.cpp:593-654 reset] */
int toast_hip_accel_reset(const void * host, size_t nbytes, const char * name);

/* Host -> device copy of the whole buffer.  [ref: accelerator.cpp:459-522 update_device] */
int toast_hip_accel_update_device(const void * host, size_t nbytes, const char * name);

/* Device -> host copy of the whole buffer; returns after the data is on the host.
 * [ref: accelerator.cpp:524-591 update_host] */
/* Upload in parts on a stream of its own (beyond the reference: its accel_update_device is one blocking copy).
 * _parts page-locks the source (a buffer not locked yet: range by range, each right before its part is enqueued, so that
 * locking part k + 1 runs while part k crosses PCIe) and returns once the copies are ENQUEUED, part k = bytes
 * [part_end[k-1], part_end[k]); _wait makes `stream` wait for part k without blocking the host; _arrived says whether part
 * k is there (no waiting); _finish blocks until all parts have arrived (call it before the source may change).  Kernels can then work on the first rows of a timestream buffer while the later
 * rows are still crossing PCIe (ops.NoiseFilter).  A pageable source falls back to the blocking copy. */
int toast_hip_accel_update_device_parts(const void * host, size_t nbytes, const char * name, const size_t * part_end,
                                        int n_parts);
int toast_hip_accel_update_device_wait(const void * host, int part, void * stream);
int toast_hip_accel_update_device_arrived(const void * host, int part, int * arrived);
int toast_hip_accel_update_device_finish(const void * host);
int toast_hip_accel_update_host(void * host, size_t nbytes, const char * name);

/* Free the device copy.  [ref: accelerator.cpp:390-457 remove] */
int toast_hip_accel_delete(const void * host, size_t nbytes, const char * name);

/* Device pointer of a registered host buffer (error if absent).
 * [ref: accelerator.hpp:115-143 device_ptr] */
int toast_hip_accel_device_ptr(const void * host, void ** device);

/* Register device memory owned by the caller (e.g. a torch tensor) as the device copy of
 * `host`; toast_hip_accel_delete() then only forgets the mapping.  Extension (no reference
 * counterpart): lets RCCL collectives run on torch tensors that the kernels also address. */
int toast_hip_accel_adopt(const void * host, size_t nbytes, void * device, const char * name);

/* Print the table of registered buffers.  [ref: accelerator.cpp:697-720 dump] */
int toast_hip_accel_dump(void);

/* Stream used by the host-pointer level (hipStream_t; NULL = default stream). */
int toast_hip_set_stream(void * stream);
/* Block until all work queued by this library on its stream has finished. */
int toast_hip_synchronize(void);

/* ------------------------------------------------------------------------------------
 * pointing_detector: quats[q_idx[d], s, :] = (flagged ? identity : boresight[s]) * fp[d]
 * [ref: ops_pointing_detector.cpp:78-227]
 *   focalplane f64[n_det,4]; boresight f64[n_samp,4]; quat_index i32[n_det];
 *   quats f64[*,n_samp,4]; shared_flags u8[n_flags] (used iff n_flags == n_samp)
 * ---------------------------------------------------------------------------------- */
int toast_hip_pointing_detector(
    const double * focalplane, const double * boresight, const int32_t * quat_index,
    int64_t n_det, double * quats, int64_t n_quat_rows, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * shared_flags, int64_t n_flags, uint8_t shared_flag_mask, int use_accel);

int toast_hip_pointing_detector_dev(
    const double * focalplane /*host*/, const double * d_boresight, const int32_t * quat_index /*host*/,
    int64_t n_det, double * d_quats, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_flags, uint8_t shared_flag_mask, void * stream);

/* ------------------------------------------------------------------------------------
 * pixels_healpix: detector quaternions -> HEALPix pixel index (NEST or RING); flagged
 * samples get -1; hit_submaps[pix / n_pix_submap] |= 1 for unflagged samples.
 * [ref: ops_pixels_healpix.cpp:1153-1417; per-sample math :44-276, :586-666]
 *   quats f64[*,n_samp,4]; pixels i64[*,n_samp]; hit_submaps u8[n_submap] (host, in/out)
 * ---------------------------------------------------------------------------------- */
int toast_hip_pixels_healpix(
    const int32_t * quat_index, int64_t n_det, const double * quats, int64_t n_quat_rows,
    const uint8_t * shared_flags, int64_t n_flags, uint8_t shared_flag_mask,
    const int32_t * pixel_index, int64_t * pixels, int64_t n_pixel_rows, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view,
    uint8_t * hit_submaps, int64_t n_submap, int64_t n_pix_submap, int64_t nside, int nest,
    int use_accel);

/* d_hit_submaps is a device array here (u8[n_submap], OR-ed in place). */
int toast_hip_pixels_healpix_dev(
    const int32_t * quat_index /*host*/, int64_t n_det, const double * d_quats,
    const uint8_t * d_shared_flags, int64_t n_flags, uint8_t shared_flag_mask,
    const int32_t * pixel_index /*host*/, int64_t * d_pixels, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view,
    uint8_t * d_hit_submaps, int64_t n_submap, int64_t n_pix_submap, int64_t nside, int nest,
    void * stream);

/* ------------------------------------------------------------------------------------
 * stokes_weights_IQU / _I
 * [ref: ops_stokes_weights.cpp:151-396 (IQU), :398-506 (I); per-sample :50-140]
 *   weights f64[*,n_samp,3] (IQU) or f64[n_det,n_samp] (I); hwp f64[n_hwp] used iff
 *   n_hwp == n_samp; epsilon, gamma, cal f64[n_det]
 * ---------------------------------------------------------------------------------- */
int toast_hip_stokes_weights_IQU(
    const int32_t * quat_index, int64_t n_det, const double * quats, int64_t n_quat_rows,
    const int32_t * weight_index, double * weights, int64_t n_weight_rows, int64_t n_samp,
    const double * hwp, int64_t n_hwp, const toast_hip_interval * intervals, int64_t n_view,
    const double * epsilon, const double * gamma, const double * cal, int iau, int use_accel);

int toast_hip_stokes_weights_IQU_dev(
    const int32_t * quat_index /*host*/, int64_t n_det, const double * d_quats,
    const int32_t * weight_index /*host*/, double * d_weights, int64_t n_samp,
    const double * d_hwp, int64_t n_hwp, const toast_hip_interval * intervals /*host*/,
    int64_t n_view, const double * epsilon /*host*/, const double * gamma /*host*/,
    const double * cal /*host*/, int iau, void * stream);

/* Mode "QU" of StokesWeights: weights f64[*,n_samp,2] = (Q, U) of the IQU triple.  The reference has
 * no kernel for it: it runs stokes_weights_IQU into a temporary [n_det,n_samp,3] host array and
 * copies two columns out (src/toast/ops/stokes_weights/stokes_weights.py:251-279). */
int toast_hip_stokes_weights_QU(
    const int32_t * quat_index, int64_t n_det, const double * quats, int64_t n_quat_rows,
    const int32_t * weight_index, double * weights, int64_t n_weight_rows, int64_t n_samp,
    const double * hwp, int64_t n_hwp, const toast_hip_interval * intervals, int64_t n_view,
    const double * epsilon, const double * gamma, const double * cal, int iau, int use_accel);

int toast_hip_stokes_weights_QU_dev(
    const int32_t * quat_index /*host*/, int64_t n_det, const double * d_quats,
    const int32_t * weight_index /*host*/, double * d_weights, int64_t n_samp,
    const double * d_hwp, int64_t n_hwp, const toast_hip_interval * intervals /*host*/,
    int64_t n_view, const double * epsilon /*host*/, const double * gamma /*host*/,
    const double * cal /*host*/, int iau, void * stream);

int toast_hip_stokes_weights_I(
    const int32_t * weight_index, int64_t n_det, double * weights, int64_t n_weight_rows,
    int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view, const double * cal,
    int use_accel);

int toast_hip_stokes_weights_I_dev(
    const int32_t * weight_index /*host*/, int64_t n_det, double * d_weights, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view, const double * cal /*host*/,
    void * stream);

/* ------------------------------------------------------------------------------------
 * scan_map (the A of A^T N^-1 A):
 *   tod[d,s] (=0 if should_zero) {+=, -=, *=} data_scale * sum_k w[d,s,k] * map[g2l[p / nps], p % nps, k]
 * for pixels p >= 0.   map_dtype selects the reference instantiation.
 * [ref: ops_scan_map.cpp:84-292 (ops_scan_map_float64/float32/int64/int32); per-sample :15-78]
 * ---------------------------------------------------------------------------------- */
#define TOAST_HIP_MAP_F64 0
#define TOAST_HIP_MAP_F32 1
#define TOAST_HIP_MAP_I64 2
#define TOAST_HIP_MAP_I32 3

int toast_hip_scan_map(
    int map_dtype, const int64_t * global2local, int64_t n_submap, int64_t n_pix_submap,
    const void * mapdata, int64_t n_local_submap, int64_t nnz,
    double * det_data, int64_t n_data_rows, const int32_t * data_index,
    const int64_t * pixels, int64_t n_pixel_rows, const int32_t * pixel_index,
    const double * weights, int64_t n_weight_rows, const int32_t * weight_index,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    double data_scale, int should_zero, int should_subtract, int should_scale, int use_accel);

/* det_weights (host, f64[n_det]) may be NULL; when given, the kernel also applies the
 * diagonal noise weight  tod *= det_weights[d]  after the scan (fused ScanMap + NoiseWeight,
 * the PCG's proj_pipe: src/toast/ops/mapmaker_solve.py:470-498). */
int toast_hip_scan_map_dev(
    int map_dtype, const int64_t * d_global2local, int64_t n_pix_submap, const void * d_mapdata,
    int64_t nnz, double * d_det_data, const int32_t * data_index /*host*/,
    const int64_t * d_pixels, const int32_t * pixel_index /*host*/,
    const double * d_weights, const int32_t * weight_index /*host*/,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals /*host*/, int64_t n_view,
    double data_scale, int should_zero, int should_subtract, int should_scale,
    const double * det_weights /*host or NULL*/, void * stream);

/* ------------------------------------------------------------------------------------
 * build_noise_weighted (the A^T N^-1 of A^T N^-1 A):
 *   zmap[g2l[p / nps], p % nps, k] += tod[d,s] * det_scale[d] * w[d,s,k]
 * for p >= 0 and unflagged samples.  [ref: ops_mapmaker_utils.cpp:93-380; per-sample :15-86]
 *   det_flags u8[*,n_flag_samp] used iff n_flag_samp == n_samp; shared_flags likewise.
 * ---------------------------------------------------------------------------------- */
int toast_hip_build_noise_weighted(
    const int64_t * global2local, int64_t n_submap, double * zmap, int64_t n_local_submap,
    int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index, const int64_t * pixels, int64_t n_pixel_rows,
    const int32_t * weight_index, const double * weights, int64_t n_weight_rows,
    const int32_t * data_index, const double * det_data, int64_t n_data_rows,
    const int32_t * flag_index, const uint8_t * det_flags, int64_t n_flag_rows, int64_t n_flag_samp,
    const double * det_scale, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, int use_accel);

int toast_hip_build_noise_weighted_dev(
    const int64_t * d_global2local, double * d_zmap, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index /*host*/, const int64_t * d_pixels,
    const int32_t * weight_index /*host*/, const double * d_weights,
    const int32_t * data_index /*host*/, const double * d_det_data,
    const int32_t * flag_index /*host*/, const uint8_t * d_det_flags, int64_t n_flag_samp,
    const double * det_scale /*host*/, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals /*host*/, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream);

/* ------------------------------------------------------------------------------------
 * noise_weight: tod[d_idx[d], s] *= detector_weights[d]   [ref: ops_noise_weight.cpp:11-119]
 * ---------------------------------------------------------------------------------- */
int toast_hip_noise_weight(
    double * det_data, int64_t n_data_rows, int64_t n_samp, const int32_t * data_index, int64_t n_det,
    const toast_hip_interval * intervals, int64_t n_view, const double * detector_weights,
    int use_accel);

int toast_hip_noise_weight_dev(
    double * d_det_data, int64_t n_samp, const int32_t * data_index /*host*/, int64_t n_det,
    const toast_hip_interval * intervals /*host*/, int64_t n_view,
    const double * detector_weights /*host*/, void * stream);

/* ------------------------------------------------------------------------------------
 * cov_apply_diag: per pixel, vec <- Sym(mat) vec with mat the packed upper triangle
 * (nnz (nnz+1)/2 values).  [ref: /root/reference/src/libtoast/src/toast_map_cov.cpp:471-528,
 * called by covariance_apply, src/toast/covariance.py:262-306]
 * ---------------------------------------------------------------------------------- */
int toast_hip_cov_apply_diag(int64_t n_sub, int64_t subsize, int64_t nnz, const double * mat,
                             double * vec, int use_accel);
int toast_hip_cov_apply_diag_dev(int64_t n_sub, int64_t subsize, int64_t nnz, const double * d_mat,
                                 double * d_vec, void * stream);

/* healpix_ang2vec / vec2ang / ang2nest / ang2ring for n angles (theta, phi) or vectors [n, 3].  sin / cos / acos /
 * atan2 come from the device math library: vectors and angles agree with the reference to rounding (tolerance class),
 * pixel numbers except for angles within an ulp of a pixel or region boundary.
 * [ref: /root/reference/src/toast/_libtoast/ops_pixels_healpix.cpp:278-349, bindings :668-815] */
int toast_hip_healpix_ang2vec(int64_t n, const double * theta, const double * phi, double * vec, int use_accel);
int toast_hip_healpix_ang2vec_dev(int64_t n, const double * d_theta, const double * d_phi, double * d_vec, void * stream);
int toast_hip_healpix_vec2ang(int64_t n, const double * vec, double * theta, double * phi, int use_accel);
int toast_hip_healpix_vec2ang_dev(int64_t n, const double * d_vec, double * d_theta, double * d_phi, void * stream);
int toast_hip_healpix_ang2pix(int64_t nside, int nest, int64_t n, const double * theta, const double * phi,
                              int64_t * pix, int use_accel);
int toast_hip_healpix_ang2pix_dev(int64_t nside, int nest, int64_t n, const double * d_theta, const double * d_phi,
                                  int64_t * d_pix, void * stream);

/* healpix_ring2nest (op 0), healpix_nest2ring (1), healpix_degrade_nest (2) / upgrade_nest (3) / degrade_ring (4) /
 * upgrade_ring (5) by `levels` powers of two, for n pixel numbers at resolution nside.
 * [ref: /root/reference/src/toast/_libtoast/ops_pixels_healpix.cpp:383-580, bindings :893-1150] */
int toast_hip_healpix_convert(int op, int64_t nside, int64_t levels, int64_t n, const int64_t * in, int64_t * out,
                              int use_accel);
int toast_hip_healpix_convert_dev(int op, int64_t nside, int64_t levels, int64_t n, const int64_t * d_in,
                                  int64_t * d_out, void * stream);

/* healpix_vec2nest / healpix_vec2ring: HEALPix pixel of n direction vectors [n, 3] (the arithmetic of the pointing
 * kernels: bit-identical to the reference).  [ref: /root/reference/src/toast/_libtoast/ops_pixels_healpix.cpp:816-893
 * -> hpix_vec2nest / hpix_vec2ring :351-381; used by src/toast/healpix.py] */
int toast_hip_healpix_vec2pix(int64_t nside, int nest, int64_t n, const double * vec, int64_t * pix, int use_accel);
int toast_hip_healpix_vec2pix_dev(int64_t nside, int nest, int64_t n, const double * d_vec, int64_t * d_pix,
                                 void * stream);

/* cov_accum_diag_hits / cov_accum_diag_invnpp: the reference's kernels behind BuildHitMap / BuildInverseCovariance at
 * the FFI level -- one stream of n_samp samples with a (local submap, pixel in submap) pair per sample (negative =
 * skipped): hits[submap * subsize + pixel] += 1;  invnpp[.. * nnz (nnz+1)/2 + (j, k >= j)] += (scale w_j) w_k.
 * [ref: /root/reference/src/toast/_libtoast/map_cov.cpp:87-197 -> src/libtoast/src/toast_map_cov.cpp:66-153; called by
 * src/toast/ops/mapmaker_utils/kernels.py:31-41].  The batched operator path is toast_hip_build_cov below. */
int toast_hip_cov_accum_diag_hits(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp, const int64_t * submap,
                                 const int64_t * subpix, int64_t * hits, int use_accel);
int toast_hip_cov_accum_diag_hits_dev(int64_t n_sub, int64_t subsize, int64_t n_samp, const int64_t * d_submap,
                                     const int64_t * d_subpix, int64_t * d_hits, void * stream);
int toast_hip_cov_accum_diag_invnpp(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp,
                                   const int64_t * submap, const int64_t * subpix, const double * weights,
                                   double scale, double * invnpp, int use_accel);
int toast_hip_cov_accum_diag_invnpp_dev(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp,
                                       const int64_t * d_submap, const int64_t * d_subpix, const double * d_weights,
                                       double scale, double * d_invnpp, void * stream);

/* cov_accum_zmap: zmap[(submap * subsize + pixel) * nnz + k] += (scale * tod) * w_k for one sample stream.
 * [ref: /root/reference/src/toast/_libtoast/map_cov.cpp:199-250 -> src/libtoast/src/toast_map_cov.cpp:204-244].
 * (cov_accum_diag = hits + invnpp + zmap of the same stream: the three calls in a row.) */
int toast_hip_cov_accum_zmap(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp, const int64_t * submap,
                            const int64_t * subpix, const double * weights, double scale, const double * tod,
                            double * zmap, int use_accel);
int toast_hip_cov_accum_zmap_dev(int64_t n_sub, int64_t subsize, int64_t nnz, int64_t n_samp, const int64_t * d_submap,
                                const int64_t * d_subpix, const double * d_weights, double scale, const double * d_tod,
                                double * d_zmap, void * stream);
/* global_to_local: (local submap, pixel in submap) of n global pixel numbers; negative pixels give (-1, -1).
 * [ref: /root/reference/src/toast/_libtoast/pixels.cpp:9-66 -> src/libtoast/include/toast/map_pixels.hpp:11-41] */
int toast_hip_global_to_local(int64_t n, const int64_t * global_pixels, int64_t n_pix_submap, const int64_t * global2local,
                              int64_t n_submap, int64_t * local_submaps, int64_t * local_pixels, int use_accel);
int toast_hip_global_to_local_dev(int64_t n, const int64_t * d_global_pixels, int64_t n_pix_submap,
                                 const int64_t * d_global2local, int64_t * d_local_submaps, int64_t * d_local_pixels,
                                 void * stream);

/* cov_mult_diag: per pixel, data1 <- packed upper triangle of Sym(data1) Sym(data2), entry (k, m >= k) taken from
 * row m, column k of the product as the reference's column-major dsymm call leaves it.
 * [ref: /root/reference/src/libtoast/src/toast_map_cov.cpp:398-469, called by covariance_multiply,
 * src/toast/covariance.py:179-221] */
int toast_hip_cov_mult_diag(int64_t n_sub, int64_t subsize, int64_t nnz, double * data1, const double * data2,
                            int use_accel);
int toast_hip_cov_mult_diag_dev(int64_t n_sub, int64_t subsize, int64_t nnz, double * d_data1, const double * d_data2,
                                void * stream);

/* ------------------------------------------------------------------------------------
 * Pixel-domain noise covariance products (used by BinMap / MapMaker setup)
 *
 * toast_hip_build_cov: mode 0 = hit map   hits[pix] += 1            (out: int64[n_loc,nps,1])
 *                      mode 1 = inverse covariance  invnpp[pix,(j,k>=j)] += w_k (w_j det_scale)
 *                                                          (out: f64[n_loc,nps,nnz(nnz+1)/2])
 *   for unflagged samples with pix >= 0 -- the operator-level semantics of BuildHitMap /
 *   BuildInverseCovariance [ref: src/toast/ops/mapmaker_utils/mapmaker_utils.py:100-210,
 *   :352-520; per-sample arithmetic src/libtoast/src/toast_map_cov.cpp:66-153].
 * toast_hip_cov_eigendecompose_diag: per-pixel rcond and (optionally) inverse of the packed
 *   symmetric blocks [ref: src/libtoast/src/toast_map_cov.cpp:246-396; binding
 *   src/toast/_libtoast/map_cov.cpp:269-325].  cond may be NULL.
 * ---------------------------------------------------------------------------------- */
int toast_hip_build_cov_dev(
    int mode, const int64_t * d_global2local, void * d_out, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index /*host*/, const int64_t * d_pixels,
    const int32_t * weight_index /*host, mode 1*/, const double * d_weights /*mode 1*/,
    const int32_t * flag_index /*host*/, const uint8_t * d_det_flags, int64_t n_flag_samp,
    const double * det_scale /*host, mode 1*/, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream);
/* BuildInverseCovariance and BuildHitMap of one CovarianceAndHits pass (src/toast/ops/mapmaker_utils.py:927-1271 runs
 * them one after the other over the same pixels and flags) in one kernel where the detector-pair merged
 * inverse-covariance kernel applies (IQU, pairing on), as two passes otherwise; same accumulation as mode 1 / mode 0 of
 * toast_hip_build_cov_dev into d_invcov [n_local_pix, nnz (nnz + 1) / 2] and d_hits [n_local_pix] (int64). */
int toast_hip_build_cov_hits_dev(
    const int64_t * d_g2l, double * d_invcov, int64_t * d_hits, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index, const double * d_weights,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale,
    uint8_t det_flag_mask, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream);
/* ... and the noise-weighted map of a timestream, zmap += A^T N^-1 d (toast_hip_build_noise_weighted_dev with `data_scale`
 * as its det_scale), from the SAME sweep: what SolveAmplitudes runs back to back over the same pixels, weights and flags --
 * CovarianceAndHits, then the right-hand side's BuildNoiseWeighted (src/toast/ops/mapmaker.py:846-900, 941-1000;
 * mapmaker_utils.py:927-1271, ops_mapmaker_utils.cpp:93-111).  Values are formed in the operation order of the separate
 * kernels; 42 instead of 33 + 41 bytes per detector-sample.  *fused = 1 when one kernel did all three (IQU, detector pairs,
 * two samples per lane: 16-byte rows, even n_samp), 0 when the call fell back to the separate sweeps (same results).
 * The flags the right-hand side would see in the reference's order -- the solver flags WITH the condition-number cut that is
 * only known after this pass -- differ from the ones seen here exactly in samples of pixels whose inverse covariance is
 * zeroed by the cut: the binned map cov x zmap is zero there either way. */
int toast_hip_build_cov_hits_signal_dev(
    const int64_t * d_g2l, double * d_invcov, int64_t * d_hits, double * d_zmap, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index, const int64_t * d_pixels, const int32_t * weight_index, const double * d_weights,
    const int32_t * data_index, const double * d_det_data, const int32_t * flag_index, const uint8_t * d_det_flags,
    int64_t n_flag_samp, const double * det_scale, const double * data_scale, uint8_t det_flag_mask, int64_t n_det,
    int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, int * fused, void * stream);
/* Host-array level of the same call (arrays resolved like toast_hip_build_cov: registered device copies with use_accel,
 * staged temporaries otherwise). */
int toast_hip_build_cov_hits(const int64_t * global2local, int64_t n_submap, double * invcov, int64_t * hits,
                             int64_t n_local_submap, int64_t n_pix_submap, int64_t nnz,
                             const int32_t * pixel_index, const int64_t * pixels, int64_t n_pixel_rows,
                             const int32_t * weight_index, const double * weights, int64_t n_weight_rows,
                             const int32_t * flag_index, const uint8_t * det_flags, int64_t n_flag_rows,
                             int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
                             int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
                             int64_t n_view, const uint8_t * shared_flags, int64_t n_shared_flags,
                             uint8_t shared_flag_mask, int use_accel);

int toast_hip_build_cov(
    int mode, const int64_t * global2local, int64_t n_submap, void * out, int64_t n_local_submap,
    int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index, const int64_t * pixels,
    int64_t n_pixel_rows, const int32_t * weight_index, const double * weights, int64_t n_weight_rows,
    const int32_t * flag_index, const uint8_t * det_flags, int64_t n_flag_rows, int64_t n_flag_samp,
    const double * det_scale, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, const uint8_t * shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, int use_accel);

int toast_hip_cov_eigendecompose_diag(int64_t n_sub, int64_t subsize, int64_t nnz, double * data,
                                      double * cond, double threshold, int invert, int use_accel);
int toast_hip_cov_eigendecompose_diag_dev(int64_t n_sub, int64_t subsize, int64_t nnz, double * d_data,
                                          double * d_cond, double threshold, int invert, void * stream);

/* ------------------------------------------------------------------------------------
 * Offset template (the M and M^T of every PCG iteration)
 * [ref: template_offset.cpp:16-147 add_to_signal, :149-332 project_signal,
 *  :334-408 apply_diag_precond]
 * ---------------------------------------------------------------------------------- */
int toast_hip_template_offset_add_to_signal(
    int64_t step_length, int64_t amp_offset, const int64_t * n_amp_views,
    const double * amplitudes, const uint8_t * amplitude_flags, int64_t n_amp,
    int32_t data_index, double * det_data, int64_t n_data_rows, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, int use_accel);

int toast_hip_template_offset_add_to_signal_dev(
    int64_t step_length, int64_t amp_offset, const int64_t * n_amp_views /*host*/,
    const double * d_amplitudes, const uint8_t * d_amplitude_flags,
    int32_t data_index, double * d_det_data, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view, void * stream);

int toast_hip_template_offset_project_signal(
    int32_t data_index, const double * det_data, int64_t n_data_rows,
    int32_t flag_index, const uint8_t * flag_data, int64_t n_flag_rows, uint8_t flag_mask,
    int64_t step_length, int64_t amp_offset, const int64_t * n_amp_views,
    double * amplitudes, const uint8_t * amplitude_flags, int64_t n_amp, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, int use_accel);

int toast_hip_template_offset_project_signal_dev(
    int32_t data_index, const double * d_det_data, int32_t flag_index, const uint8_t * d_flag_data,
    uint8_t flag_mask, int64_t step_length, int64_t amp_offset, const int64_t * n_amp_views /*host*/,
    double * d_amplitudes, const uint8_t * d_amplitude_flags, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view, void * stream);

/* Batched (all detectors in one launch) forms of the two per-detector offset kernels, and the
 * fused left-hand side of the PCG for offset templates.  Extensions: the reference calls the
 * per-detector bindings in a Python loop (src/toast/templates/offset/offset.py:727-881) and runs
 * add_to_signal / build_noise_weighted / scan_map / noise_weight / project_signal as separate
 * passes (src/toast/ops/mapmaker_solve.py:342-506).
 *   offset_accumulate :   zmap += A^T N^-1 (M a)
 *   offset_scan_project : a_out += M^T [ det_w (M a - A map) ]  over samples with clear flags
 * amp_offsets[d] = first amplitude of detector d (Offset._det_start).  flag_index / d_flag_data
 * may be NULL (no sample flags). */
int toast_hip_template_offset_add_to_signal_multi_dev(
    int64_t step_length, const int64_t * amp_offsets /*host*/, const int64_t * n_amp_views /*host*/,
    const double * d_amplitudes, const uint8_t * d_amplitude_flags, const int32_t * data_index /*host*/,
    int64_t n_det, double * d_det_data, int64_t n_samp, const toast_hip_interval * intervals /*host*/,
    int64_t n_view, void * stream);

int toast_hip_template_offset_project_signal_multi_dev(
    const int32_t * data_index /*host*/, const double * d_det_data, const int32_t * flag_index /*host*/,
    const uint8_t * d_flag_data, uint8_t flag_mask, int64_t step_length, const int64_t * amp_offsets /*host*/,
    const int64_t * n_amp_views /*host*/, double * d_amplitudes, const uint8_t * d_amplitude_flags,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals /*host*/, int64_t n_view,
    void * stream);

int toast_hip_offset_accumulate_dev(
    int64_t step_length, const int64_t * amp_offsets /*host*/, const int64_t * n_amp_views /*host*/,
    const double * d_amplitudes, const uint8_t * d_amplitude_flags, const int64_t * d_global2local,
    double * d_zmap, int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index /*host*/,
    const int64_t * d_pixels, const int32_t * weight_index /*host*/, const double * d_weights,
    const int32_t * flag_index /*host*/, const uint8_t * d_det_flags, int64_t n_flag_samp,
    const double * det_scale /*host*/, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream);

/* The map-maker's last two steps in one pass: zmap += A^T N^-1 (d - M a) -- the template-cleaned signal binned without
 * being written -- the reference's ApplyAmplitudes(op = "subtract") and the accumulation of the final BinMap
 * (src/toast/ops/mapmaker.py:531-608, mapmaker_templates.py:1205-1261) with the per-sample arithmetic of offset
 * add_to_signal (template_offset.cpp:16-122: the template value is 0 + a for an unflagged amplitude, 0 otherwise), the
 * subtraction d - template, and build_noise_weighted (ops_mapmaker_utils.cpp:15-86).  d = row data_index[k] of d_signal
 * (only read).  nnz = 3, an even number of samples per row and 16-byte aligned rows (what cached IQU pointing has); other
 * shapes are refused and the caller runs the two operators.  Other arguments as toast_hip_offset_accumulate_dev. */
int toast_hip_offset_clean_accumulate_dev(
    int64_t step_length, const int64_t * amp_offsets /*host*/, const int64_t * n_amp_views /*host*/,
    const double * d_amplitudes, const uint8_t * d_amplitude_flags, const int64_t * d_global2local,
    double * d_zmap, int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index /*host*/,
    const int64_t * d_pixels, const int32_t * weight_index /*host*/, const double * d_weights,
    const int32_t * data_index /*host*/, const double * d_signal, const int32_t * flag_index /*host*/,
    const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale /*host*/, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals /*host*/, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream);

int toast_hip_offset_scan_project_dev(
    int64_t step_length, const int64_t * amp_offsets /*host*/, const int64_t * n_amp_views /*host*/,
    const double * d_amplitudes_in, double * d_amplitudes_out, const uint8_t * d_amplitude_flags,
    const int64_t * d_global2local, const double * d_map, int64_t n_pix_submap, int64_t nnz,
    const int32_t * pixel_index /*host*/, const int64_t * d_pixels, const int32_t * weight_index /*host*/,
    const double * d_weights, const int32_t * flag_index /*host or NULL*/, const uint8_t * d_flag_data,
    uint8_t flag_mask, const double * det_weights /*host*/, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view, void * stream);
/* The solver's right-hand side tail in one pass: a_out += M^T N^-1 (d - A z) for the timestreams d (row signal_index[d]
 * of the device buffer d_signal, which is only read) -- the reference's Copy, ScanMap(subtract), NoiseWeight and
 * TemplateMatrix(transpose) of SolverRHS (src/toast/ops/mapmaker_solve.py:140-229) with the per-sample arithmetic of
 * scan_map (ops_scan_map.cpp:15-78), noise_weight (ops_noise_weight.cpp:14-40) and offset project_signal
 * (template_offset.cpp:149-331).  Other arguments as toast_hip_offset_scan_project_dev. */
int toast_hip_offset_scan_project_signal_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views, const int32_t * signal_index,
    const double * d_signal, double * d_amplitudes_out, const uint8_t * d_amplitude_flags, const int64_t * d_g2l,
    const double * d_map, int64_t n_pix_submap, int64_t nnz, const int32_t * pixel_index, const int64_t * d_pixels,
    const int32_t * weight_index, const double * d_weights, const int32_t * flag_index, const uint8_t * d_flag_data,
    uint8_t flag_mask, const double * det_weights, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, void * stream);

/* ------------------------------------------------------------------------------------
 * The solver's pointing cache in 20 bytes per detector-sample (csrc/packed_pointing.hip).  Every PCG iteration sweeps
 * the cached pointing twice [ref: SolverLHS._exec, src/toast/ops/mapmaker_solve.py:342-506] and reads pixel (int64),
 * three weights and a flag byte per sample: 33 B.  During a solve none of it changes, the intensity weight is a
 * per-detector constant [ref: stokes_weights_IQU, src/toast/_libtoast/ops_stokes_weights.cpp:77-140: weights[0] = cal],
 * the kernels only need the OFFSET of the pixel in the local map, and the flags only say "skip".
 *   pack_pointing:  d_key[n_det][n_samp] (uint32: local pixel offset + 1 in bits 0-29, 0 = no pixel of the local map;
 *       bit 30 = flagged for the accumulation: (det_flags & det_flag_mask) | (shared_flags & shared_flag_mask); bit 31 =
 *       flagged for the projection: proj_flags & proj_flag_mask), d_qu[n_det][n_samp][2] (the Q and U weights),
 *       d_cal[n_det] (the intensity weight).  Rows follow the order of the index arrays.  *packable = 1 when the
 *       intensity weight is the same number in every sample in view of each row and every offset fits 30 bits
 *       (Nside <= 8192); otherwise 0, and the caller keeps using toast_hip_offset_accumulate_dev / _scan_project_dev.
 *       Waits for the stream (once per solve).  "Absent" flag arrays as elsewhere: length != n_samp.
 *   offset_accumulate_packed / offset_scan_project_packed: the two sweeps of toast_hip_offset_accumulate_dev /
 *       toast_hip_offset_scan_project_dev (nnz = 3) from the packed cache: the same products in the same order (the
 *       projection bit for bit; the accumulation up to the order of its atomic additions, as between any two runs).
 *       n_samp must be even.
 *   pair words (pair_words != NULL in pack_pointing, returned 1): the two detectors of a co-pointing pair -- rows 2b and
 *       2b + 1 -- see the same pixel in every sample, so when every pair of the call agrees on its offsets (checked) and
 *       they fit 28 bits, row 2b of d_key is rewritten as ONE word per pair-sample (offset + 1 in bits 0-27,
 *       accumulation / projection flag of member e in bits 28 + 2 e / 29 + 2 e; row 2b + 1 is then unused) and the
 *       sweeps, told so by their pair_words argument, read 4 + 2 x 16 B per pair-sample = 18 B per detector-sample, with
 *       one map gather per pair-sample in the projection.  toast_hip_offset_pack_pairs_dev is that step on its own, for
 *       callers that pack their rows in several calls (pair_words = NULL there) and merge at the end.
 * ---------------------------------------------------------------------------------- */
int toast_hip_offset_pack_pointing_dev(
    const int64_t * d_g2l, int64_t n_pix_submap, const int32_t * pixel_index, const int64_t * d_pixels,
    const int32_t * weight_index, const double * d_weights, const int32_t * acc_flag_index, const uint8_t * d_det_flags,
    int64_t n_flag_samp, uint8_t det_flag_mask, const uint8_t * d_shared_flags, int64_t n_shared_flags,
    uint8_t shared_flag_mask, const int32_t * proj_flag_index, const uint8_t * d_proj_flags, int64_t n_proj_flag_samp,
    uint8_t proj_flag_mask, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    uint32_t * d_key, double * d_qu, double * d_cal, int * packable, int * pair_words, void * stream);
/* The three calls above in ONE sweep over the cached pointing when the rows come in co-pointing pairs: pair words, both rows
 * of Q / U weights and the float2 pair weight sums d_corr[(n_det + 1) / 2][n_samp] straight from the pixels, weights and
 * flags (33 B read, 22 B written per detector-sample instead of four passes).  *pair_weights = 1: d_corr is usable (at most
 * one marked component in a hundred).  Pairs that do not share their pixels in every sample, an odd n_samp or d_corr == NULL:
 * the separate passes run instead (*pair_weights = 0).  Same outputs bit for bit, except that row 2 b + 1 of d_key -- which
 * the pair-word sweeps never read -- is not written.  TOAST_HIP_PACK_ONEPASS=0: always the separate passes.
 * [ref: what is packed is the pointing SolverLHS._exec sweeps, src/toast/ops/mapmaker_solve.py:342-506] */
int toast_hip_offset_pack_pointing_onepass_dev(
    const int64_t * d_g2l, int64_t n_pix_submap, const int32_t * pixel_index /*host*/, const int64_t * d_pixels,
    const int32_t * weight_index /*host*/, const double * d_weights, const int32_t * acc_flag_index /*host*/,
    const uint8_t * d_det_flags, int64_t n_flag_samp, uint8_t det_flag_mask, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, const int32_t * proj_flag_index /*host*/,
    const uint8_t * d_proj_flags, int64_t n_proj_flag_samp, uint8_t proj_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view, uint32_t * d_key, double * d_qu, double * d_cal,
    float * d_corr, int * packable, int * pair_words, int * pair_weights, void * stream);
int toast_hip_offset_pack_pairs_dev(uint32_t * d_key, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
                                    int64_t n_view, int * pair_words, void * stream);
/* Pair weights (with pair words): the Q / U weights of an orthogonal pair are negatives of each other up to a few ulps,
 * so q_a + q_b is exact, fits a float exactly and gives q_b = (q_a + q_b) - q_a back exactly.  Checked per sample in view
 * and per component; where it does not hold (weights that are rounding noise around zero, NaN at a pole) the component
 * gets a NaN marker and the sweeps read that partner weight from d_qu's partner row, which stays in place: lossless for
 * every input.  *ok = 0 when more than one component in a hundred is marked (pairs of different calibration).
 * d_corr[(n_det + 1) / 2][n_samp][2] float.  The sweeps then read 14 instead of 18 B per detector-sample: pass d_corr as
 * d_pair_corr below (NULL: both rows of d_qu are read). */
int toast_hip_offset_pack_pair_weights_dev(const double * d_qu, float * d_corr, int64_t n_det, int64_t n_samp,
                                           const toast_hip_interval * intervals, int64_t n_view, int * ok, void * stream);
int toast_hip_offset_accumulate_packed_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views, const double * d_amplitudes,
    const uint8_t * d_amplitude_flags, double * d_zmap, const uint32_t * d_key, const double * d_qu, const double * d_cal,
    const double * det_scale, int pair_words, const float * d_pair_corr, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals, int64_t n_view, void * stream);
int toast_hip_offset_scan_project_packed_dev(
    int64_t step_length, const int64_t * amp_offsets, const int64_t * n_amp_views, const double * d_amps_in,
    double * d_amps_out, const uint8_t * d_amplitude_flags, const double * d_map, const uint32_t * d_key,
    const double * d_qu, const double * d_cal, const double * det_weights, int pair_words, const float * d_pair_corr,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view, void * stream);

int toast_hip_template_offset_apply_diag_precond(
    const double * offset_var, const double * amp_in, const uint8_t * amplitude_flags,
    double * amp_out, int64_t n_amp, int use_accel);

int toast_hip_template_offset_apply_diag_precond_dev(
    const double * d_offset_var, const double * d_amp_in, const uint8_t * d_amplitude_flags,
    double * d_amp_out, int64_t n_amp, void * stream);

/* Offset template noise prior and its preconditioners on device-resident amplitudes.  The
 * reference applies them on the host only and refuses use_accel
 * (src/toast/templates/offset/offset.py:884-960 `_add_prior`, :963-1005 `_apply_precond`); these
 * two entry points are what a maintainer binds there instead of the NotImplementedError.
 *
 * The local amplitudes are a concatenation of n_seg segments, one per (detector, observation,
 * view) in the template's own order: d_seg_start[n_seg + 1] holds their first amplitudes and the
 * total.  Every table argument is a device pointer.
 *
 * toast_hip_template_offset_convolve_dev: out (+)= scipy.signal.convolve(in, filter, "same") per
 * segment with the segment's filter (d_filters + d_filt_start[s], d_filt_len[s] taps, odd),
 * then out = 0 where the amplitude is flagged.  accumulate != 0 is `_add_prior`
 * (offset.py:918-943), accumulate == 0 the Toeplitz preconditioner of precond_width <= 1
 * (offset.py:981-989).  in and out must differ.  max_segment_len / max_filter_len (host values:
 * the longest segment and filter) select the launch geometry -- filters longer than 32 taps take
 * an LDS-tiled, register-blocked kernel.
 *
 * toast_hip_template_offset_banded_solve_dev: out = cho_solve_banded((factor, lower=True), in)
 * per segment (offset.py:990-1001), out = 0 where flagged.  The factor of segment s, band width
 * w = d_band_width[s] <= max_band_width <= 256, is handed over as two [n_amp_view][w] row-major
 * tables at d_forward / d_backward + d_band_start[s], with cb[k][j] = L[j + k][j] scipy's lower
 * banded storage:
 *   forward[i][0] = backward[i][0] = 1 / cb[0][i]
 *   forward[i][k]  = cb[k][i]      = L[i + k][i]  (0 when i + k >= n),  k = 1 .. w-1
 *   backward[i][k] = cb[k][i - k]  = L[i][i - k]  (0 when i - k < 0),   k = 1 .. w-1
 * i.e. row i of each table couples unknown i to the unknowns solved AFTER it in that sweep (the
 * column-oriented substitution of LAPACK's dtbsv). */
int toast_hip_template_offset_convolve_dev(
    int64_t n_amp, int64_t n_seg, const int64_t * d_seg_start, int64_t max_segment_len,
    const int64_t * d_filt_start, const int64_t * d_filt_len, int64_t max_filter_len,
    const double * d_filters, const double * d_amp_in, const uint8_t * d_amplitude_flags,
    double * d_amp_out, int accumulate, void * stream);

int toast_hip_template_offset_banded_solve_dev(
    int64_t n_seg, const int64_t * d_seg_start, const int32_t * d_band_width, int32_t max_band_width,
    const int64_t * d_band_start, const double * d_forward, const double * d_backward,
    const double * d_amp_in, const uint8_t * d_amplitude_flags, double * d_amp_out, void * stream);

/* toast_hip_template_offset_banded_cholesky_dev: factorise, per segment, the preconditioner matrix
 *   M = diag(d_diag_scale[s] / d_offset_var[i]) + Toeplitz(band of segment s)
 * (offset.py:522-545; scipy.linalg.cholesky_banded(lower=True) on the host there) and write the
 * factor directly in the two table layouts described above.  The Toeplitz band of segment s is
 * d_toeplitz[d_toeplitz_start[s] .. + d_toeplitz_len[s]), zero beyond; an amplitude with
 * variance 0 (flagged) gets no diagonal term.  d_backward must be zero-filled by the caller
 * (entries that leave the segment are not written).  d_status[s] != 0: not positive definite at
 * this band width (the caller widens the band, offset.py:546-566).  Band widths up to 64. */
int toast_hip_template_offset_banded_cholesky_dev(
    int64_t n_seg, const int64_t * d_seg_start, const int32_t * d_band_width, int32_t max_band_width,
    const int64_t * d_band_start, const int64_t * d_toeplitz_start, const int32_t * d_toeplitz_len,
    const double * d_toeplitz, const double * d_diag_scale, const double * d_offset_var,
    double * d_forward, double * d_backward, int32_t * d_status, void * stream);

/* ------------------------------------------------------------------------------------
 * Template regression for toast.ops.GroundFilter (device-resident buffers; host index arrays)
 *
 * The reference fits, per detector, a set of templates shared by all detectors of an observation
 * (Legendre trend, Legendre / binned functions of azimuth) with four host kernels of
 * src/libtoast/src/toast_tod_filter.cpp -- `legendre` :269-331, `bin_proj` :160-177, `bin_invcov`
 * :179-215, `add_templates` :333-355 (bindings src/toast/_libtoast/tod_filter.cpp:100, 213, 254,
 * 291) -- called in a Python loop over detectors (src/toast/ops/groundfilter.py:443-500).
 *
 * toast_hip_legendre_templates_dev: templates[order - start_order][i] = normalised Legendre
 *   polynomial of d_x[i], orders [start_order, stop_order); same recurrence and rounding.
 * toast_hip_template_fit_dev: for every detector d (rows signal_index[d] / flag_index[d]),
 *   good[d][i] = !(shared_flags[i] & shared_flag_mask) && !(det_flags[d][i] & det_flag_mask):
 *     d_proj[d][r]            = sum_i T[r][i] signal[d][i] good[d][i]              (bin_proj)
 *     d_gram_common[r][c]     = sum_i T[r][i] T[c][i] over samples with good shared flags
 *     d_gram_flagged[d][r][c] = the same sum over samples with good shared flags that detector d flags
 *     d_n_flagged[d]          = the number of those samples
 *   so that bin_invcov of detector d is d_gram_common - d_gram_flagged[d].  All four outputs are
 *   overwritten.  d_det_flags / d_shared_flags may be NULL (no flags of that kind).
 * toast_hip_template_subtract_dev: signal[d][i] -= sum_{r >= first_template} coeff[d][r] T[r][i]
 *   for ALL samples, the sum accumulated from zero in template order (add_templates into a zeroed
 *   buffer, then `ref -= fit`: groundfilter.py:384-393).  d_coeff is [n_det][n_template].
 * ---------------------------------------------------------------------------------- */
int toast_hip_legendre_templates_dev(const double * d_x, int64_t n_samp, int64_t start_order,
                                     int64_t stop_order, double * d_templates, void * stream);

/* out[i] = keep ? (d_src ? d_src[i] : 1.0) : 0.0 with keep = (d_key[i] == value) if keep_equal else
 * (d_key[i] != value): the split (left / right going scans) and binned-azimuth templates of
 * groundfilter.py:208-257 from a per-sample key (scan direction, azimuth bin). */
int toast_hip_template_select_dev(const double * d_src, const int32_t * d_key, int32_t value, int keep_equal,
                                  int64_t n_samp, double * d_out, void * stream);

int toast_hip_template_fit_dev(
    const double * d_templates, int64_t n_template, int64_t n_samp, const int32_t * signal_index /*host*/,
    const double * d_signal, const int32_t * flag_index /*host*/, const uint8_t * d_det_flags,
    uint8_t det_flag_mask, const uint8_t * d_shared_flags, uint8_t shared_flag_mask, int64_t n_det,
    double * d_proj, double * d_gram_common, double * d_gram_flagged, int64_t * d_n_flagged, void * stream);

int toast_hip_template_subtract_dev(
    const double * d_templates, int64_t n_template, int64_t first_template, int64_t n_samp,
    const int32_t * signal_index /*host*/, double * d_signal, const double * d_coeff, int64_t n_det,
    void * stream);

/* ------------------------------------------------------------------------------------
 * FFT noise weighting
 *
 * toast_hip_fft_convolve: in-place convolution (or deconvolution) of each selected timestream
 * with a Fourier-domain kernel -- toast.fft.convolve(..., algorithm="numpy")
 * [ref: src/toast/fft.py:163-212, 252-350; caller src/toast/ops/noise_filter.py:130-188].
 * The kernel is passed the way the reference evaluates it: |K| and arg K as PCHIP piecewise
 * cubics on the common kernel frequencies (`knots`), as scipy PPoly coefficients
 * coef[kernel][interval][4] (highest power first); ang_coef may be NULL (real kernel >= 0).
 * n_kernel is 1 (one kernel for all detectors) or n_det.  `apodize` is the n_reflect-sample
 * half window of fft.py:163-171.  n_fft = toast_hip_fft_length(n_samp).
 *
 * toast_hip_fft_r1d: batched 1-D real transforms in FFTW half-complex layout -- the
 * FFTPlanReal1D exec of the reference [ref: src/libtoast/include/toast/math_fft.hpp:24-82,
 * src/libtoast/src/toast_math_fft_fftw.cpp:26-128, cuFFT variant toast_math_fft_cufft.cpp:16-238]:
 * forward: out = scale * r2hc(in); backward: out = (scale / length) * hc2r(in).
 * ---------------------------------------------------------------------------------- */
int64_t toast_hip_fft_length(int64_t n_samp);
/* Which implementation toast_hip_fft_convolve* uses for this timestream length: 1 = the fused
 * three-pass kernels (toast_amd/csrc/fft_fused.hip: padding / apodisation, real-FFT packing, the
 * kernel multiplication and the crop fused into two column passes and one row pass of a four-step
 * FFT; power-of-two n_fft >= 8192), 0 = the rocFFT pipeline (short transforms, or every length
 * with TOAST_HIP_FFT=rocfft in the environment).  Both give the same result to ~1e-15 relative. */
int toast_hip_fft_fused(int64_t n_samp);
/* Override the choice at run time: rocfft_only != 0 = the rocFFT pipeline for every length,
 * 0 = automatic (what TOAST_HIP_FFT=rocfft / unset select at start-up). */
void toast_hip_fft_select(int rocfft_only);
/* Widths of the kernels' impulse responses, as toast.fft.convolve measures them before growing the flagged regions
 * (src/toast/fft.py:836-872: an impulse of 100 at sample n_samp / 2 of an empty timestream through the same
 * convolution; from the peak of |response| walk both ways while it exceeds 2 % of the peak): extents[d] = imax - imin
 * for d < n_det (n_kernel == 1: one common response).  Kernel arguments as toast_hip_fft_convolve; everything runs on
 * the device, `extents` is a host array.  stream NULL = the manager's stream. */
int toast_hip_fft_impulse_extents(int64_t n_det, int64_t n_samp, double rate, const double * knots, int64_t n_knot,
                                  const double * mag_coef, const double * ang_coef, int64_t n_kernel,
                                  int deconvolve, const double * apodize, int64_t n_apodize, int32_t * extents,
                                  void * stream);
/* Flag bookkeeping of toast.fft.convolve / NoiseFilter on the device: for detector d (row flag_index[d] of the uint8
 * [n_flag_rows, n_samp] buffer) grow every run of samples with (flag & mask) != 0 by extents[d] samples on both sides
 * exactly as the reference's extend_flags does (src/toast/utils.py:1055-1113: the mask is ASSIGNED, the last sample is
 * never assigned), and with edges != 0 OR the mask into the first and last extents[d] samples
 * (src/toast/fft.py:935-945).  or_row (host array [n_samp], may be NULL) is OR-ed into every selected row first: the
 * shared flags NoiseFilter merges into the detector flags before the convolution (src/toast/ops/noise_filter.py:118-126).
 * Host buffer, or its registered device copy with use_accel != 0. */
int toast_hip_fft_extend_flags(uint8_t * flags, int64_t n_flag_rows, const int32_t * flag_index, int64_t n_det,
                               int64_t n_samp, uint8_t mask, const int32_t * extents, int edges,
                               const uint8_t * or_row, int use_accel);
/* Points per thread (16 or 8; anything else = default) of the row pass, the forward and the inverse
 * column pass of the fused kernels: 16 = 256-thread workgroups with radix-16 stages, 8 = 512-thread
 * workgroups with radix-8 stages and twice the waves per SIMD.  Defaults 8 / 8 / 8; start-up value
 * from TOAST_HIP_FFT_POINTS="rows,cols_fwd,cols_inv". */
void toast_hip_fft_points(int rows, int cols_fwd, int cols_inv);
/* Row pass of the fused kernels: 2 (default) = the tile in registers and the LDS only an exchange buffer, 16 points per
 * lane: a row of N2 = 2048 bins in two waves, the row pair (k1, N1 - k1) in one workgroup, three workgroups per CU
 * (csrc/fft_reg.hip); 3 = 32 points per lane, one row per wave, two workgroups per CU; 0 = both rows of a pair interleaved
 * in one 64 KB LDS tile; 1 = one row per 32 KB LDS tile (experiment: register bound, slower).  Start-up value from
 * TOAST_HIP_FFT_ROWS=reg|reg32|lds|split.  Same results to rounding. */
void toast_hip_fft_rows_split(int split);
/* Column passes of the fused kernels: 1 (default) = the tile in registers (csrc/fft_reg.hip) for n_fft 2^22 and 2^23
 * (N2 = 2048, even n_samp / padding: 128- and 64-byte pieces where the LDS tile gives 64 and 32), 2 = for n_fft 2^21 too,
 * 0 = the tile in LDS (k_fft_cols) at every length.  Start-up value from TOAST_HIP_FFT_COLS=reg|reg9|lds.  Same results
 * to rounding. */
void toast_hip_fft_cols_reg(int on);
/* Row length of the four-step factorisation M = N1 x N2 of the fused kernels: 2048 (64 KB row-pair tiles, two
 * workgroups per CU in the row pass, 128-byte pieces in the column passes) or 1024 (32 KB tiles, four workgroups per
 * CU, 64-byte pieces; only for M <= 2^20).  Start-up value from TOAST_HIP_FFT_N2.  Same results to rounding. */
void toast_hip_fft_rows_n2(int n2);
/* Host only, no device needed: the order in which the forward column pass of the fused kernels takes its n_tiles column
 * tiles (cols_per_tile complex columns each) so that the tiles reading the same 128-byte lines of the timestream -- directly
 * and as the two mirror images of the padded series, set_rfft_input, src/toast/fft.py:163-188 -- run on one XCD (workgroup
 * b runs on XCD b mod 8 and takes tile order[b]).  Returns n_tiles and fills order[0 .. n_tiles), or 0 when the geometry
 * (n_samp, n_buffer, n_reflect) does not split into classes that fill the 8 XCDs evenly: the pass then keeps contiguous
 * column ranges per XCD.  TOAST_HIP_FFT_XCD=0|1 switches it off. */
int toast_hip_fft_mirror_tile_order(int64_t n_samp, int64_t n_buffer, int64_t n_reflect, int64_t n_tiles,
                                    int64_t cols_per_tile, int32_t * order);
/* HBM bytes per timestream sample that the passes of that implementation move (accounting for
 * bench.py / DESIGN.md, not a measurement). */
double toast_hip_fft_pipeline_bytes(int64_t n_samp);

int toast_hip_fft_convolve(
    double * det_data, int64_t n_data_rows, const int32_t * data_index, int64_t n_det, int64_t n_samp,
    double rate, const double * knots, int64_t n_knot, const double * mag_coef, const double * ang_coef,
    int64_t n_kernel, int deconvolve, const double * apodize, int64_t n_apodize, int use_accel);

int toast_hip_fft_convolve_dev(
    double * d_det_data, const int32_t * data_index /*host*/, int64_t n_det, int64_t n_samp, double rate,
    const double * knots /*host*/, int64_t n_knot, const double * mag_coef /*host*/,
    const double * ang_coef /*host or NULL*/, int64_t n_kernel, int deconvolve,
    const double * apodize /*host*/, int64_t n_apodize, int64_t max_batch, void * stream);

int toast_hip_fft_r1d(int forward, int64_t length, int64_t count, const double * in, double * out,
                      double scale, int use_accel);
int toast_hip_fft_r1d_dev(int forward, int64_t length, int64_t count, const double * d_in, double * d_out,
                          double scale, void * stream);

/* ------------------------------------------------------------------------------------
 * Deterministic debug mode (TOAST_HIP_DETERMINISTIC=1 in the environment, or this switch).
 * The production A^T kernels add run-reduced partial sums with fp64 atomics, so zmap / the
 * inverse covariance differ from run to run in the last bits.  With the mode on,
 * toast_hip_build_noise_weighted* and toast_hip_build_cov* (mode 1) sum every pixel's
 * contributions in (detector, interval, sample) order -- the order of the reference's host path
 * [ref: src/toast/_libtoast/ops_mapmaker_utils.cpp:294-378, src/libtoast/src/toast_map_cov.cpp:96-153]
 * -- by a stable sort of (pixel, det-sample) pairs and a sequential segmented sum
 * (toast_amd/csrc/deterministic.hip): bit-identical between runs and to the reference's host
 * result.  ~20x slower than the atomic kernels.  Also switched: toast_hip_template_offset_project_signal*
 * (one thread per amplitude, samples in increasing order) and toast_hip_vec_dot_dev (block partials
 * summed in block order).  The fused / on-the-fly accumulate kernels are not covered (the operators
 * avoid them while the mode is on).
 * ---------------------------------------------------------------------------------- */
int toast_hip_set_deterministic(int on);
int toast_hip_get_deterministic(void);

/* Stokes weights within rounding of a pole: the reference's -sqrt(1 - z*z) is NaN when z*z rounds
 * above 1, and with it the Q / U weights [ref: src/toast/_libtoast/ops_stokes_weights.cpp:66-75].
 * The library reproduces the reference's NaNs sample for sample (the default since round 4: same results on the
 * same inputs); on == 0 (or TOAST_HIP_STOKES_REFERENCE_NAN=0) selects the device formulation that is finite there
 * (weights of modulus eta * cal). */
int toast_hip_set_stokes_reference_nan(int on);

/* ScanMask on device copies: det_flags[d,s] |= flag_value where mask[g2l[p/nps], p%nps] &
 * mask_bits (mask: u8[n_local_submap, n_pix_submap, 1]).  Operator-level semantics of the
 * reference's host-only ScanMask [ref: src/toast/ops/scan_map/scan_map.py:283-320]. */
int toast_hip_scan_mask_dev(const int64_t * d_global2local, const uint8_t * d_mask, int64_t n_pix_submap,
                            uint8_t mask_bits, uint8_t flag_value, const int32_t * pixel_index /*host*/,
                            const int64_t * d_pixels, const int32_t * flag_index /*host*/,
                            uint8_t * d_det_flags, int64_t n_det, int64_t n_samp,
                            const toast_hip_interval * intervals /*host*/, int64_t n_view, void * stream);

/* Device-to-device copy on the stream (Copy operator on resident buffers). */
int toast_hip_copy_dev(void * d_dst, const void * d_src, size_t nbytes, void * stream);
int toast_hip_memset_dev(void * d_dst, int value, size_t nbytes, void * stream);
/* n_block blocks of block_bytes: block dst_index[i] of d_dst <- block src_index[i] of d_src (host index arrays).  The
 * device form of PixelData.sync_alltoallv for ranks that hold DIFFERENT local submaps moves the local submaps into the
 * union of all ranks' submaps with it, and back [ref: src/toast/pixels.py:792-967]. */
int toast_hip_block_move_dev(void * d_dst, const void * d_src, int64_t n_block, int64_t block_bytes,
                             const int64_t * dst_index /*host*/, const int64_t * src_index /*host*/, void * stream);

/* ------------------------------------------------------------------------------------
 * Pointing on the fly: the accumulate / scan kernels evaluate
 * pointing_detector -> pixels_healpix -> stokes_weights per det-sample in registers instead of
 * reading cached pixels / weights.  This is the fused form of the reference's
 * full_pointing=False pipelines [ref: src/toast/ops/mapmaker_binning.py:265-271,
 * src/toast/ops/mapmaker_solve.py:476-489], which re-run the three pointing operators inside
 * every pass.  The descriptor collects the arguments of those three operators
 * [ref: ops_pointing_detector.cpp:78-88, ops_pixels_healpix.cpp:1153-1167,
 * ops_stokes_weights.cpp:151-163]; d_* members are device pointers, the others host arrays of
 * n_det entries (row i = detector i of the call).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    const double * d_boresight;     /* [n_samp, 4] */
    const uint8_t * d_shared_flags; /* [n_shared_flags]; used when n_shared_flags == n_samp */
    int64_t n_shared_flags;
    uint8_t shared_flag_mask;
    const double * focalplane;      /* host [n_det, 4] detector quaternions */
    const double * d_hwp;           /* [n_hwp]; HWP angle used when n_hwp == n_samp (nnz 3 only) */
    int64_t n_hwp;
    const double * epsilon;         /* host [n_det] or NULL (= 0) */
    const double * gamma;           /* host [n_det] or NULL (= 0) */
    const double * cal;             /* host [n_det] or NULL (= 1) */
    int IAU;
    int64_t nside;
    int nest;
    int nnz;                        /* 1: intensity weights, 3: IQU */
    /* Optional compact pixel cache (toast_hip_compact_pixels_dev): when not NULL the kernels read
     * the int32 LOCAL map index of each det-sample from row compact_index[i] of this [rows, n_samp]
     * array (4 B/det-sample) and evaluate only the Stokes weights on the fly. */
    const int32_t * d_compact_pixels;
    const int32_t * compact_index;  /* host [n_det] */
    /* Optional [n_hwp, 2] table (cos 4 hwp, sin 4 hwp) from toast_hip_hwp_table_dev: the kernels
     * then read 16 B per time sample (shared by all detectors) instead of evaluating sincos. */
    const double * d_hwp_table;
} toast_hip_otf_pointing;

/* Compact pixel cache: local map index  global2local[pix / n_pix_submap] * n_pix_submap +
 * pix % n_pix_submap  as int32 (-1 for pix < 0 or a submap that is not local) for every sample in
 * the intervals.  n_local_submap * n_pix_submap must fit in int32. */
int toast_hip_compact_pixels_dev(const int64_t * d_g2l, int64_t n_pix_submap, int64_t n_local_submap,
                                 const int32_t * pixel_index, const int64_t * d_pixels,
                                 const int32_t * compact_index, int32_t * d_compact_pixels, int64_t n_det,
                                 int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
                                 void * stream);

/* Quaternion-free pointing expansion: the results of pointing_detector + pixels_healpix
 * (int64 pixels, -1 for flagged boresight samples, hit_submaps marked on the device) and of
 * pointing_detector + stokes_weights_IQU / _I ([rows, n_samp, nnz] weights) written straight
 * from the boresight; same values as the separate kernels. */
int toast_hip_otf_pixels_healpix_dev(const toast_hip_otf_pointing * pointing, const int32_t * pixel_index,
                                     int64_t * d_pixels, int64_t n_det, int64_t n_samp,
                                     const toast_hip_interval * intervals, int64_t n_view,
                                     uint8_t * d_hit_submaps, int64_t n_submap, int64_t n_pix_submap,
                                     void * stream);
int toast_hip_otf_stokes_weights_dev(const toast_hip_otf_pointing * pointing, const int32_t * weight_index,
                                     double * d_weights, int64_t n_det, int64_t n_samp,
                                     const toast_hip_interval * intervals, int64_t n_view, void * stream);

/* table[i] = (cos(4 hwp[i]), sin(4 hwp[i])): the HWP modulation 2 (2 (gamma - hwp)) of
 * stokes_weights_IQU [ref: ops_stokes_weights.cpp:96-99] by angle addition from per-detector
 * cos / sin 4 gamma; d_table is 16-byte aligned, 2 n_samp doubles. */
int toast_hip_hwp_table_dev(const double * d_hwp, int64_t n_samp, double * d_table, void * stream);

/* The same cache computed straight from the boresight (pointing_detector -> pixels_healpix ->
 * global2local in registers): no int64 pixel buffer is needed at all.  `pointing`'s weight
 * members and d_compact_pixels are ignored. */
int toast_hip_otf_compact_pixels_dev(const toast_hip_otf_pointing * pointing, const int64_t * d_g2l,
                                     int64_t n_pix_submap, int64_t n_local_submap,
                                     const int32_t * compact_index, int32_t * d_compact_pixels, int64_t n_det,
                                     int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
                                     void * stream);

/* zmap += P^T N^-1 d  with on-the-fly pointing (== pointing chain + build_noise_weighted). */
int toast_hip_otf_build_noise_weighted_dev(
    const toast_hip_otf_pointing * pointing, const int64_t * d_g2l, double * d_zmap, int64_t n_pix_submap,
    const int32_t * data_index, const double * d_det_data, const int32_t * flag_index,
    const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream);

/* d = (zero ? 0 : d) -/+ data_scale * P m, optionally followed by d *= det_weights[det]
 * (== pointing chain + scan_map(float64) [+ noise_weight]); det_weights may be NULL. */
int toast_hip_otf_scan_map_dev(const toast_hip_otf_pointing * pointing, const int64_t * d_g2l,
                               const double * d_map, int64_t n_pix_submap, double * d_det_data,
                               const int32_t * data_index, int64_t n_det, int64_t n_samp,
                               const toast_hip_interval * intervals, int64_t n_view, double data_scale,
                               int should_zero, int should_subtract, const double * det_weights,
                               void * stream);

/* Fused PCG left-hand side halves for Offset templates with on-the-fly pointing (see
 * toast_hip_offset_accumulate_dev / toast_hip_offset_scan_project_dev). */
int toast_hip_otf_offset_accumulate_dev(
    const toast_hip_otf_pointing * pointing, int64_t step_length, const int64_t * amp_offsets,
    const int64_t * n_amp_views, const double * d_amplitudes, const uint8_t * d_amplitude_flags,
    const int64_t * d_g2l, double * d_zmap, int64_t n_pix_submap, const int32_t * flag_index,
    const uint8_t * d_det_flags, int64_t n_flag_samp, const double * det_scale, uint8_t det_flag_mask,
    int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals, int64_t n_view,
    const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream);
/* zmap += P^T N^-1 (d - M a) with on-the-fly pointing: ApplyAmplitudes(op = "subtract") + the accumulation of the final
 * BinMap [ref: src/toast/ops/mapmaker.py:531-608] in one pass, the cleaned timestream never written (the cached-pointing
 * form: toast_hip_offset_clean_accumulate_dev).  d = row data_index[k] of d_signal (only read). */
int toast_hip_otf_offset_clean_accumulate_dev(
    const toast_hip_otf_pointing * pointing, int64_t step_length, const int64_t * amp_offsets /*host*/,
    const int64_t * n_amp_views /*host*/, const double * d_amplitudes, const uint8_t * d_amplitude_flags,
    const int64_t * d_g2l, double * d_zmap, int64_t n_pix_submap, const int32_t * data_index /*host*/,
    const double * d_signal, const int32_t * flag_index /*host*/, const uint8_t * d_det_flags, int64_t n_flag_samp,
    const double * det_scale /*host*/, uint8_t det_flag_mask, int64_t n_det, int64_t n_samp,
    const toast_hip_interval * intervals /*host*/, int64_t n_view, const uint8_t * d_shared_flags,
    int64_t n_shared_flags, uint8_t shared_flag_mask, void * stream);
int toast_hip_otf_offset_scan_project_dev(
    const toast_hip_otf_pointing * pointing, int64_t step_length, const int64_t * amp_offsets,
    const int64_t * n_amp_views, const double * d_amplitudes_in, double * d_amplitudes_out,
    const uint8_t * d_amplitude_flags, const int64_t * d_g2l, const double * d_map, int64_t n_pix_submap,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp, uint8_t det_flag_mask,
    const double * det_weights, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
    int64_t n_view, void * stream);
/* The same with the timestreams as the signal (row signal_index[d] of d_signal, only read): the tail of SolverRHS for
 * uncached pointing, see toast_hip_offset_scan_project_signal_dev. */
int toast_hip_otf_offset_scan_project_signal_dev(
    const toast_hip_otf_pointing * pointing, int64_t step_length, const int64_t * amp_offsets,
    const int64_t * n_amp_views, const int32_t * signal_index, const double * d_signal, double * d_amplitudes_out,
    const uint8_t * d_amplitude_flags, const int64_t * d_g2l, const double * d_map, int64_t n_pix_submap,
    const int32_t * flag_index, const uint8_t * d_det_flags, int64_t n_flag_samp, uint8_t det_flag_mask,
    const double * det_weights, int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
    int64_t n_view, void * stream);

/* counts[amplitude] += number of samples under that offset amplitude with (det_flags & flag_mask) != 0
 * (as doubles; the caller zeroes d_counts): the input of the good-fraction cut and of the
 * preconditioner variances in Offset initialisation [ref: src/toast/templates/offset/offset.py:262-343]. */
int toast_hip_offset_count_flagged_dev(int64_t step_length, const int64_t * amp_offsets,
                                       const int64_t * n_amp_views, double * d_counts, const int32_t * flag_index,
                                       const uint8_t * d_det_flags, uint8_t flag_mask, int64_t n_det, int64_t n_samp,
                                       const toast_hip_interval * intervals, int64_t n_view, void * stream);
/* d_mask[i] |= bit where d_value[i] < threshold (device arrays of n entries): the pixels whose inverse condition
 * number is below the solver's cut, SolveAmplitudes' rcond mask (src/toast/ops/mapmaker_templates.py:902-939). */
int toast_hip_threshold_mask_dev(int64_t n, const double * d_value, double threshold, uint8_t bit, uint8_t * d_mask,
                                 void * stream);
/* Offset template set-up, second half (src/toast/templates/offset.py:300-343): flag and variance of every baseline from
 * the flagged-sample counts of toast_hip_offset_count_flagged_dev.  Detector d owns amplitudes amp_offsets[d] + j,
 * j < n_len, with amp_len[j] samples each and noise weight det_weight[d] (host arrays).  Cut (flag 1, variance 0) when
 * n_good / amp_len <= good_fraction, det_weight <= 0 or amp_len == 0; otherwise variance = 1 / (det_weight * n_good).
 * d_n_bad, d_amp_flags, d_variance: device arrays over all amplitudes. */
int toast_hip_offset_variance_dev(int64_t n_det, int64_t n_len, const int64_t * amp_offsets, const double * det_weight,
                                  const int64_t * amp_len, const double * d_n_bad, double good_fraction,
                                  uint8_t * d_amp_flags, double * d_variance, void * stream);

/* Solver flags [ref: src/toast/ops/mapmaker_templates.py:764-810]: for the samples inside the
 * intervals  out[out_index[d]][s] = ((det_flags[flag_index[d]][s] & det_flag_mask) != 0) |
 * ((shared_flags[s] & shared_flag_mask) != 0);  the optional inputs are absent when their length
 * differs from n_samp.  When outside_value >= 0 the whole [n_out_rows, n_samp] buffer is first
 * set to it (samples outside every interval keep that value), otherwise it is left untouched. */
int toast_hip_combine_flags_dev(uint8_t * d_out, const int32_t * out_index, const uint8_t * d_det_flags,
                                int64_t n_flag_samp, const int32_t * flag_index, uint8_t det_flag_mask,
                                const uint8_t * d_shared_flags, int64_t n_shared_flags, uint8_t shared_flag_mask,
                                int64_t n_det, int64_t n_samp, const toast_hip_interval * intervals,
                                int64_t n_view, int64_t n_out_rows, int outside_value, void * stream);

/* PCG vector algebra on device-resident amplitude vectors
 * [ref: src/toast/templates/amplitudes.py:400-565]:  y = a x + b y  (b == 0 overwrites), and the
 * flagged dot product  sum_i x_i y_i over entries with both flags clear (flag pointers may be
 * NULL); the dot returns its value to the host (synchronises the stream). */
int toast_hip_vec_axpby_dev(int64_t n, double a, const double * d_x, double b, double * d_y, void * stream);
int toast_hip_vec_dot_dev(int64_t n, const double * d_x, const double * d_y, const uint8_t * d_flags_x,
                          const uint8_t * d_flags_y, double * result /*host*/, void * stream);

/* ------------------------------------------------------------------------------------
 * PCG with its scalars on the device [ref: src/toast/ops/mapmaker_solve.py:524-755, the recurrence and its
 * convergence (relative < convergence or sqsum < 1e-30), stall (every 10 iterations: last_best < 2 sqsum_best) and
 * iteration-limit tests].  The state (alpha, beta, delta, the residual norms, iteration counter, history of relative
 * residuals) lives in a device block of toast_hip_pcg_state_bytes(); dot products reduce into it, one-thread stage
 * kernels do the scalar arithmetic, the vector updates read alpha / beta from it: the host only enqueues and reads the
 * status `lag` iterations late.  After the solver has finished (done != 0) further stages / updates change nothing.
 *   one iteration:  lhs_out = A proposal;  dot(proposal, lhs_out); stage 1;  result += alpha proposal (axpby ALPHA, ONE);
 *       residual -= alpha lhs_out (NEG_ALPHA, ONE);  dot(residual, residual); stage 2;  precond = M^-1 residual;
 *       dot(precond, residual); stage 3;  proposal = live precond + beta proposal (LIVE, BETA).
 * `allreduce` != 0 sums the dot product over the ranks of toast_hip_comm before the stage reads it.
 * ---------------------------------------------------------------------------------- */
enum { TOAST_HIP_PCG_RUNNING = 0, TOAST_HIP_PCG_CONVERGED = 1, TOAST_HIP_PCG_STALLED = 2, TOAST_HIP_PCG_NOT_FINITE = 3,
       TOAST_HIP_PCG_MAX_ITER = 4 };
enum { TOAST_HIP_PCG_ONE = 0, TOAST_HIP_PCG_ALPHA = 1, TOAST_HIP_PCG_NEG_ALPHA = 2, TOAST_HIP_PCG_BETA = 3,
       TOAST_HIP_PCG_LIVE = 4 };
typedef struct toast_hip_pcg_status {
    int64_t iteration;   /* iterations completed */
    int64_t done;        /* TOAST_HIP_PCG_* */
    int64_t n_history;   /* relative residuals recorded */
    double relative;     /* the last of them */
    double sqsum;        /* the last residual norm squared */
} toast_hip_pcg_status;
int toast_hip_pcg_state_bytes(int64_t n_iter_max, size_t * bytes);
int toast_hip_pcg_init_dev(void * d_state, double sqsum_init, double delta, double convergence, int64_t n_iter_min,
                           int64_t n_iter_max, void * stream);
/* state.tmp (+)= sum_i x_i y_i over entries with both flags clear (flag pointers may be NULL); stage != 0 runs that
 * stage right behind the reduction in the same launch (one process), stage = 0 leaves it to toast_hip_pcg_stage_dev
 * (several processes: the sum over the ranks comes in between). */
int toast_hip_pcg_dot_dev(void * d_state, int64_t n, const double * d_x, const double * d_y, const uint8_t * d_flags_x,
                          const uint8_t * d_flags_y, int accumulate, int stage, void * stream);
int toast_hip_pcg_stage_dev(void * d_state, int stage, int allreduce, void * stream);
/* result += alpha proposal;  residual -= alpha lhs_out  (one launch) */
int toast_hip_pcg_step_dev(const void * d_state, int64_t n, const double * d_proposal, double * d_result,
                           const double * d_lhs_out, double * d_residual, void * stream);
/* The two vector updates whose OUTPUT a dot product of the recurrence reads, fused with that dot product (same
 * accumulate / stage arguments as toast_hip_pcg_dot_dev; same bits as the separate calls -- the element -> thread
 * assignment and the summation order are shared):
 *   step_dot:          result += alpha proposal;  residual -= alpha lhs_out;  state.tmp (+)= residual . residual
 *                      [ref: src/toast/ops/mapmaker_solve.py:679-690]
 *   precond_diag_dot:  out = residual * var where the residual's flag is clear, else 0 (template_offset_apply_diag_precond
 *                      [ref: src/toast/_libtoast/template_offset.cpp:334-396]);  state.tmp (+)= out . residual
 *                      [ref: src/toast/ops/mapmaker_solve.py:726-737] */
int toast_hip_pcg_step_dot_dev(void * d_state, int64_t n, const double * d_proposal, double * d_result,
                               const double * d_lhs_out, double * d_residual, const uint8_t * d_flags, int accumulate,
                               int stage, void * stream);
int toast_hip_pcg_precond_diag_dot_dev(void * d_state, int64_t n, const double * d_var, const double * d_residual,
                                       const uint8_t * d_flags_residual, double * d_out, const uint8_t * d_flags_out,
                                       int accumulate, int stage, void * stream);
/* y = S[a_sel] x + S[b_sel] y with S = the state's scalars (TOAST_HIP_PCG_ONE ... _LIVE) */
int toast_hip_pcg_axpby_dev(const void * d_state, int64_t n, int a_sel, const double * d_x, int b_sel, double * d_y,
                            void * stream);
/* Enqueue an asynchronous copy of the status and return the one enqueued `lag` calls ago (0 = this one: waits for the
 * stream; 1 = the host runs one iteration ahead of the device); zeros while fewer than `lag` calls have been made.
 * The lagged copies go through one ring of page-locked slots per process, owned by the state initialised last; a call
 * for any other state block waits for the stream and returns that block's current status. */
int toast_hip_pcg_status_dev(void * d_state, int lag, toast_hip_pcg_status * out, void * stream);
/* Waits for the stream; copies min(n_history, capacity) relative residuals and the final status to the host. */
int toast_hip_pcg_history_dev(void * d_state, double * history, int64_t capacity, toast_hip_pcg_status * final_status,
                              void * stream);

/* ------------------------------------------------------------------------------------
 * Multi-GPU: the process' RCCL communicator (one process per GPU), collectives enqueued on the
 * caller's stream -- kernel -> collective -> kernel is stream order, no host synchronisation.
 * Replaces the reference's host-side MPI reductions of the pixel-domain objects:
 *   PixelData.sync_allreduce  [ref: src/toast/pixels.py:710-780]  (D2H, MPI Allreduce in 10 MB pieces, H2D:
 *                              src/toast/ops/mapmaker_utils/mapmaker_utils.py:885-925)
 *   PixelData.sync_alltoallv(local_func)  [ref: src/toast/pixels.py:942-967] and its users
 *   covariance_invert / _multiply / _apply(use_alltoallv=True)  [ref: src/toast/covariance.py:34-131, 134-221, 224-306]
 * librccl is opened at run time by the first call (a single-GPU process does not need it);
 * TOAST_HIP_RCCL_LIB=<path> names the library to open instead of searching for one.
 * The 128-byte unique id is created on one rank and handed to the others by the host side's own
 * transport (MPI bcast in TOAST, torch.distributed here) before toast_hip_comm_init.
 * ---------------------------------------------------------------------------------- */
#define TOAST_HIP_COMM_ID_BYTES 128
enum { TOAST_HIP_COMM_F64 = 0, TOAST_HIP_COMM_F32 = 1, TOAST_HIP_COMM_I64 = 2, TOAST_HIP_COMM_I32 = 3, TOAST_HIP_COMM_U8 = 4 };
enum { TOAST_HIP_COMM_SUM = 0, TOAST_HIP_COMM_MAX = 1, TOAST_HIP_COMM_MIN = 2 };
/* 1 when librccl can be opened with every entry point needed (no communicator is created): lets the ranks agree
 * before the collective toast_hip_comm_init -- a rank that cannot load RCCL would leave the others waiting in it. */
int toast_hip_comm_available(void);
int toast_hip_comm_unique_id(void * id128);
int toast_hip_comm_init(const void * id128, int n_ranks, int rank);
/* n_ranks = 0 / rank = -1 when there is no communicator; rccl_version as ncclGetVersion reports it */
int toast_hip_comm_info(int * n_ranks, int * rank, int * rccl_version);
int toast_hip_comm_destroy(void);
/* in place on d_buf: every rank ends with the reduction over all ranks */
int toast_hip_comm_allreduce_dev(void * d_buf, int64_t count, int dtype, int op, void * stream);
int toast_hip_comm_broadcast_dev(void * d_buf, int64_t count, int dtype, int root, void * stream);
/* d_send holds n_ranks * recv_count elements; rank r receives the reduction of piece r */
int toast_hip_comm_reduce_scatter_dev(const void * d_send, void * d_recv, int64_t recv_count, int dtype, int op,
                                      void * stream);
/* d_recv holds n_ranks * send_count elements; piece r = rank r's d_send */
int toast_hip_comm_all_gather_dev(const void * d_send, void * d_recv, int64_t send_count, int dtype, void * stream);
/* Owner-computes on pixel-domain objects that every rank holds with the same local submaps: rank r owns the pixels
 * [r * per, (r + 1) * per) of the n_px local pixels, per = ceil(n_px / n_ranks) (toast_hip_comm_pixel_shard).
 *   map_reduce_apply: map <- [cov .] (reduce ? sum over ranks of map : map) -- reduce-scatter (owners receive the sum),
 *       cov_apply_diag on the owned shard (skipped when d_cov is NULL), all-gather (results go back).  With d_cov = NULL
 *       and reduce = 1 this is sync_alltoallv() with the default local_func (sum).
 *   cov_invert / cov_mult: every rank works on its shard of the replicated matrices, the shards are gathered. */
int toast_hip_comm_pixel_shard(int64_t n_px, int64_t * first, int64_t * count);
/* the same rule for any (n_ranks, rank), no communicator needed: pixels [first, first + count), per_rank = ceil(n_px / n_ranks) */
int toast_hip_comm_shard_of(int64_t n_px, int n_ranks, int rank, int64_t * first, int64_t * count, int64_t * per_rank);
int toast_hip_comm_map_reduce_apply_dev(int64_t n_px, int64_t nnz, const double * d_cov, double * d_map, int reduce,
                                        void * stream);
/* How toast_hip_comm_map_reduce_apply_dev does its work (TOAST_HIP_COMM_MODE sets the start-up value): "owner" (default:
 * reduce-scatter, multiplication on the owned shard, all-gather on the caller's stream), "allreduce" (one all-reduce, every
 * rank multiplies the whole map: the reference's sync_allreduce + covariance_apply, pixels.py:710-780), "peer" (at most 16
 * ranks of one node: no RCCL on the data path -- every rank writes the foreign slices of its map into their owners' exchange
 * buffers, opened through hipIpc handles, owners add them in rank order and multiply, every rank reads the finished slices
 * back; all xGMI links of the mesh carry 1/N of the map at once, RCCL provides the two barriers; fails with
 * TOAST_HIP_ERR_DEVICE on every rank if the buffers cannot be opened), "peer:flags" (the same with the barriers done by
 * device flags in each other's uncached memory: no library call per reduction; a wait gives up after
 * TOAST_HIP_COMM_PEER_TIMEOUT_MS, the map of that reduction starts with NaN, and the error is raised by the rank's next
 * reduction, toast_hip_comm_check, toast_hip_comm_set_mode or toast_hip_comm_destroy, whichever comes first).  In the
 * "peer" modes toast_hip_comm_allreduce_dev sums maps of doubles (>= 4096 values per rank) through the exchange buffers
 * too.  Same results in every mode (to the rounding of the sums' order).  Collective: every rank must use the same mode.
 * (Round 4's "sliced:S" is gone: two streams of one communicator do not overlap its collectives.) */
int toast_hip_comm_set_mode(const char * mode);
/* Bytes per lane and access of the "peer" exchange kernels: 8 (default: system-scope atomic loads / stores of one double)
 * or 16 (ordinary non-temporal 16-byte accesses; used when the slices are even and the map 16-byte aligned).  Start-up
 * value from TOAST_HIP_COMM_PEER_WIDTH.  Same values either way. */
int toast_hip_comm_set_peer_width(int bytes);
/* Synchronises the stream and raises a pending "peer:flags" time-out (0: nothing pending). */
int toast_hip_comm_check(void * stream);
/* mode "peer": reductions done through the exchange buffers, times the buffers were (re-)established (collective hipIpc
 * exchange; grows with the largest map), bytes of this rank's exchange buffer now */
int toast_hip_comm_peer_stats(int64_t * reductions, int64_t * establishments, int64_t * exchange_bytes);
/* mode "peer": is the exchange buffer fine-grained memory?  TOAST_HIP_COMM_PEER_MEM=fine (hipDeviceMallocFinegrained:
 * coherent between the GPUs by construction) | coarse (default: an ordinary hipMalloc; visibility rests on the kernels'
 * system-scope accesses / fences and the kernel boundaries around the barriers) -- read when the buffers are established.
 * [the reference has no device-to-device path: its sync_allreduce goes through host MPI, pixels.py:710-780] */
int toast_hip_comm_peer_mem(int * fine_grained);
int toast_hip_comm_get_mode(char * mode, size_t len);
int toast_hip_comm_cov_invert_dev(int64_t n_px, int64_t nnz, double * d_cov, double * d_rcond, double threshold,
                                  int invert, void * stream);
int toast_hip_comm_cov_mult_dev(int64_t n_px, int64_t nnz, double * d_cov1, const double * d_cov2, void * stream);

/* ------------------------------------------------------------------------------------
 * Test / measurement helpers (device primitives compared per operation with the CPU).
 * ---------------------------------------------------------------------------------- */
int toast_hip_test_math_dev(int op /*0 atan2(a,b), 1 sqrt(a), 2 a/b*/, int64_t n, const double * d_a,
                            const double * d_b, double * d_out, void * stream);

#ifdef __cplusplus
}
#endif
#endif /* TOAST_HIP_H */
