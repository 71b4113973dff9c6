// Text in, text out: FASTA ingest on the device (A1 + A2: record bodies -> class indices, where the records of a file start) and the
// TSV rows of the result (A12, host code).  Byte and integer work: every function is bit-exact against the reference's Python.
#include "dgrp_common.h"
#include "scan.h"
#include <vector>
#include <algorithm>
#include <string.h>

// ------------------------------------------------------------------------------------------
// A1 + A2 fused on the device: FASTA record body -> class indices (see include/deepgrp_hip.h).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fasta_class_of(uint32_t c)
{
    uint32_t l = c | 0x20u;
    return l == 'a' ? 0u : l == 'c' ? 1u : l == 'g' ? 2u : l == 't' ? 3u : 4u;
}

// per tile: kept-byte count; globally: "not plain" flag.  g[0] = bad flag, g[1] = first non-N compact
// index (atomicMin), g[2] = last non-N compact index + 1 (atomicMax)
__global__ void __launch_bounds__(256) fasta_count_kernel(const uint8_t *__restrict__ raw, int64_t n, uint64_t *__restrict__ tilecnt,
                                                          unsigned long long *__restrict__ g)
{
    __shared__ uint64_t lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    uint64_t c = 0;
    bool bad = false;
    for (int j = 0; j < 8; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) {
            const uint32_t b = raw[i];
            const bool lineend = b == '\n' || b == '\r';
            c += lineend ? 0 : 1;
            if (b >= 128 || (b <= 32 && !lineend)) bad = true;                      // non-ASCII / other whitespace
            if (b == '\r' && (i + 1 >= n || raw[i + 1] != '\n')) bad = true;         // lone CR: a line break in text mode
            if (b == '\n') {
                if (i == 0) bad = true;                                             // body starts with a blank line
                if (i + 1 < n && raw[i + 1] == '\n') bad = true;                     // blank line
                if (i + 2 < n && raw[i + 1] == '\r' && raw[i + 2] == '\n') bad = true;
            }
            if (b == '\r' && i == 0) bad = true;
        }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&g[0], 1ull);
    __syncthreads();
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ void __launch_bounds__(256) fasta_scatter_kernel(const uint8_t *__restrict__ raw, int64_t n,
                                                            const uint64_t *__restrict__ tileoff, uint8_t *__restrict__ idx,
                                                            unsigned long long *__restrict__ g)
{
    __shared__ uint64_t lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;
    uint32_t b[8];
    uint64_t s = 0;
    for (int j = 0; j < 8; ++j) {
        b[j] = base + j < n ? raw[base + j] : (uint32_t)'\n';
        s += (b[j] == '\n' || b[j] == '\r') ? 0 : 1;
    }
    uint64_t ex = block_exclusive_scan(s, nullptr, lds) + tileoff[blockIdx.x];
    long long first = 0x7fffffffffffffffll, last = -1;
    for (int j = 0; j < 8; ++j) {
        if (b[j] == '\n' || b[j] == '\r') continue;
        idx[ex] = (uint8_t)fasta_class_of(b[j]);
        if ((b[j] | 0x20u) != 'n') {                        // upper() precedes the N stripping in the reference
            if ((long long)ex < first) first = (long long)ex;
            last = (long long)ex + 1;
        }
        ++ex;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long f2 = __shfl_xor(first, o), l2 = __shfl_xor(last, o);
        first = f2 < first ? f2 : first;
        last = l2 > last ? l2 : last;
    }
    // One candidate per workgroup, and only where it can still move the record's bounds: every wave sending its own pair was half
    // a million same-address atomics per 250 MB -- 11.3 ms for a kernel that streams its bytes in a fraction of a millisecond.  (A
    // stale read of the current bound is safe: the bounds only tighten, so a candidate that does not beat an older value cannot
    // beat the current one.)
    __shared__ long long red[8];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = first; red[4 + (threadIdx.x >> 6)] = last; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            first = red[w] < first ? red[w] : first;
            last = red[4 + w] > last ? red[4 + w] : last;
        }
        if (last >= 0) {
            const volatile unsigned long long *gv = g;
            if ((unsigned long long)first < gv[1]) atomicMin((unsigned long long *)&g[1], (unsigned long long)first);
            if ((unsigned long long)last > gv[2]) atomicMax((unsigned long long *)&g[2], (unsigned long long)last);
        }
    }
}

DGRP_EXPORT int64_t dgrp_fasta_workspace_bytes(int64_t nbytes)
{
    if (nbytes < 0) return 0;
    return dgrp_align_up(((nbytes + SCAN_TILE - 1) / SCAN_TILE + 2) * 8, 256) + 256;
}

DGRP_EXPORT int dgrp_fasta_encode(const uint8_t *d_raw, int64_t nbytes, uint8_t *d_idx, int64_t *h_info, void *d_work,
                                  int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(nbytes >= 0 && h_info, "dgrp_fasta_encode: bad arguments");
    h_info[0] = 1; h_info[1] = 0; h_info[2] = 0; h_info[3] = 0;
    if (nbytes == 0) return DGRP_OK;
    DGRP_REQUIRE(d_raw && d_idx && d_work, "dgrp_fasta_encode: NULL pointer");
    if (work_bytes < dgrp_fasta_workspace_bytes(nbytes)) {
        dgrp_set_error("dgrp_fasta_encode: workspace too small");
        return DGRP_ENOMEM;
    }
    const int64_t ntiles = (nbytes + SCAN_TILE - 1) / SCAN_TILE;
    uint64_t *tiles = (uint64_t *)d_work;
    uint64_t *grand = tiles + ntiles + 1;                      // [0] scan total
    unsigned long long *g = (unsigned long long *)((char *)d_work + dgrp_fasta_workspace_bytes(nbytes) - 256);
    const unsigned long long init[3] = { 0ull, 0x7fffffffffffffffull, 0ull };
    DGRP_HIP(hipMemcpyAsync(g, init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(fasta_count_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, d_raw, nbytes, tiles, g);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, stream, tiles, ntiles, grand);
    DGRP_LAUNCH_CHECK();
    hipLaunchKernelGGL(fasta_scatter_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, d_raw, nbytes, tiles, d_idx, g);
    DGRP_LAUNCH_CHECK();
    unsigned long long hg[3];
    uint64_t total = 0;
    DGRP_HIP(hipMemcpyAsync(hg, g, sizeof(hg), hipMemcpyDeviceToHost, stream));
    DGRP_HIP(hipMemcpyAsync(&total, grand, 8, hipMemcpyDeviceToHost, stream));
    DGRP_HIP(hipStreamSynchronize(stream));
    h_info[0] = hg[0] ? 0 : 1;
    h_info[1] = (int64_t)total;
    if (hg[2] == 0) {                                          // no non-N character at all
        // sequence.pyx:27-30: startpos runs to the end, length runs to 0 -> negative unless empty
        h_info[2] = (int64_t)total;
        h_info[3] = -(int64_t)total;
    } else {
        h_info[2] = (int64_t)hg[1];
        h_info[3] = (int64_t)hg[2] - (int64_t)hg[1];
    }
    return DGRP_OK;
}

// One workgroup per record body: count, running offset and scatter tile after tile (a short record is a few tiles;
// no separate scan, no launch per record).  g = [bad flag, first non-N compact index, last non-N compact index + 1,
// kept characters] per record (32 words apart).
__global__ void __launch_bounds__(256) fasta_record_kernel(const uint8_t *__restrict__ raw_base, const int64_t *__restrict__ off,
                                                           const int64_t *__restrict__ len, const uint8_t *__restrict__ small,
                                                           uint8_t *__restrict__ idx_base, unsigned long long *__restrict__ gbase)
{
    __shared__ uint64_t lds[4];
    const int64_t r = blockIdx.x;
    if (!small[r]) return;
    const int64_t n = len[r];
    const uint8_t *raw = raw_base + off[r];
    uint8_t *idx = idx_base + off[r];
    unsigned long long *g = gbase + r * 32;
    uint64_t running = 0;
    bool bad = false;
    long long first = 0x7fffffffffffffffll, last = -1;
    for (int64_t t0 = 0; t0 < n; t0 += SCAN_TILE) {
        const int64_t base = t0 + (int64_t)threadIdx.x * 8;
        uint32_t b[8];
        uint64_t c = 0;
        for (int j = 0; j < 8; ++j) {
            const int64_t i = base + j;
            b[j] = i < n ? raw[i] : (uint32_t)'\n';
            if (i < n) {
                const bool lineend = b[j] == '\n' || b[j] == '\r';
                c += lineend ? 0 : 1;
                if (b[j] >= 128 || (b[j] <= 32 && !lineend)) bad = true;
                if (b[j] == '\r' && (i + 1 >= n || raw[i + 1] != '\n')) bad = true;
                if (b[j] == '\n') {
                    if (i == 0) bad = true;
                    if (i + 1 < n && raw[i + 1] == '\n') bad = true;
                    if (i + 2 < n && raw[i + 1] == '\r' && raw[i + 2] == '\n') bad = true;
                }
                if (b[j] == '\r' && i == 0) bad = true;
            }
        }
        uint64_t tot = 0;
        uint64_t ex = block_exclusive_scan(c, &tot, lds) + running;
        running += tot;
        for (int j = 0; j < 8; ++j) {
            if (b[j] == '\n' || b[j] == '\r') continue;
            idx[ex] = (uint8_t)fasta_class_of(b[j]);
            if ((b[j] | 0x20u) != 'n') {
                if ((long long)ex < first) first = (long long)ex;
                last = (long long)ex + 1;
            }
            ++ex;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long f2 = __shfl_xor(first, o), l2 = __shfl_xor(last, o);
        first = f2 < first ? f2 : first;
        last = l2 > last ? l2 : last;
    }
    const bool anybad = __any(bad);
    if ((threadIdx.x & 63) == 0) {
        if (anybad) atomicOr(&g[0], 1ull);
        if (last >= 0) {
            atomicMin(&g[1], (unsigned long long)first);
            atomicMax(&g[2], (unsigned long long)last);
        }
    }
    if (threadIdx.x == 0) g[3] = running;
}

// Many record bodies of ONE uploaded buffer in a single call: the same three kernels per record, queued back to
// back, one read-back and one synchronisation for all of them (a file of thousands of short records would
// otherwise pay an upload and a wait per record).  Record r is the byte range [h_off[r], h_off[r] + h_len[r]) of
// d_raw; its class indices go to d_idx + h_off[r] (same offsets, capacity h_len[r]); h_info gets 4 values per
// record as dgrp_fasta_encode defines them.  Workspace: dgrp_fasta_batch_workspace_bytes(nrec, total bytes).
DGRP_EXPORT int64_t dgrp_fasta_batch_workspace_bytes(int64_t nrec, int64_t total_bytes)
{
    if (nrec < 0 || total_bytes < 0) return 0;
    // per record: the tiles of its own scan (rounded up) + 3 words, and 256 B of counters
    return dgrp_align_up(((total_bytes + SCAN_TILE - 1) / SCAN_TILE + 4 * nrec + 8) * 8, 256) + nrec * 256 + dgrp_align_up(nrec * 17, 256) + 256;
}

DGRP_EXPORT int dgrp_fasta_encode_batch(const uint8_t *d_raw, int64_t nrec, const int64_t *h_off, const int64_t *h_len,
                                        uint8_t *d_idx, int64_t *h_info, void *d_work, int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(nrec >= 0 && (nrec == 0 || (h_off && h_len && h_info)), "dgrp_fasta_encode_batch: bad arguments");
    if (nrec == 0) return DGRP_OK;
    int64_t total = 0;
    for (int64_t r = 0; r < nrec; ++r) {
        DGRP_REQUIRE(h_off[r] >= 0 && h_len[r] >= 0, "dgrp_fasta_encode_batch: negative range");
        total += h_len[r];
    }
    DGRP_REQUIRE(total == 0 || (d_raw && d_idx && d_work), "dgrp_fasta_encode_batch: NULL pointer");
    if (work_bytes < dgrp_fasta_batch_workspace_bytes(nrec, total)) {
        dgrp_set_error("dgrp_fasta_encode_batch: workspace too small");
        return DGRP_ENOMEM;
    }
    const int64_t tile_words = (total + SCAN_TILE - 1) / SCAN_TILE + 4 * nrec + 8;     // sum of (tiles + 3) per record
    uint64_t *tiles_base = (uint64_t *)d_work;
    unsigned long long *gbase = (unsigned long long *)((char *)d_work + dgrp_align_up(tile_words * 8, 256));     // 32 words per record
    std::vector<unsigned long long> init((size_t)nrec * 32, 0ull);
    for (int64_t r = 0; r < nrec; ++r) init[(size_t)r * 32 + 1] = 0x7fffffffffffffffull;
    DGRP_HIP(hipMemcpyAsync(gbase, init.data(), init.size() * 8, hipMemcpyHostToDevice, stream));
    // records up to 1 MiB: one workgroup each in ONE launch; larger ones: the three kernels per record
    const int64_t SMALL_BYTES = 1 << 20;
    unsigned char *tab = (unsigned char *)(gbase + nrec * 32);           // off[nrec], len[nrec] (int64), small[nrec] (bytes)
    int64_t *d_off = (int64_t *)tab, *d_len = d_off + nrec;
    uint8_t *d_small = (uint8_t *)(d_len + nrec);
    std::vector<uint8_t> small((size_t)nrec);
    bool any_small = false;
    for (int64_t r = 0; r < nrec; ++r) { small[(size_t)r] = h_len[r] > 0 && h_len[r] <= SMALL_BYTES; any_small |= small[(size_t)r] != 0; }
    if (any_small) {
        DGRP_HIP(hipMemcpyAsync(d_off, h_off, (size_t)nrec * 8, hipMemcpyHostToDevice, stream));
        DGRP_HIP(hipMemcpyAsync(d_len, h_len, (size_t)nrec * 8, hipMemcpyHostToDevice, stream));
        DGRP_HIP(hipMemcpyAsync(d_small, small.data(), (size_t)nrec, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(fasta_record_kernel, dim3((unsigned)nrec), dim3(256), 0, stream, d_raw, d_off, d_len, d_small, d_idx, gbase);
    }
    std::vector<int64_t> grand_at((size_t)nrec, -1);
    int64_t word = 0;
    for (int64_t r = 0; r < nrec; ++r) {
        const int64_t nbytes = h_len[r];
        if (nbytes == 0 || small[(size_t)r]) continue;
        const int64_t ntiles = (nbytes + SCAN_TILE - 1) / SCAN_TILE;
        uint64_t *tiles = tiles_base + word;
        uint64_t *grand = tiles + ntiles + 1;
        grand_at[(size_t)r] = word + ntiles + 1;
        word += ntiles + 3;
        unsigned long long *g = gbase + r * 32;
        const uint8_t *raw = d_raw + h_off[r];
        hipLaunchKernelGGL(fasta_count_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, raw, nbytes, tiles, g);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, stream, tiles, ntiles, grand);
        hipLaunchKernelGGL(fasta_scatter_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, raw, nbytes, tiles,
                           d_idx + h_off[r], g);
    }
    DGRP_LAUNCH_CHECK();
    std::vector<unsigned long long> hg((size_t)nrec * 32);
    std::vector<uint64_t> htiles((size_t)(word > 0 ? word : 1));
    DGRP_HIP(hipMemcpyAsync(hg.data(), gbase, hg.size() * 8, hipMemcpyDeviceToHost, stream));
    if (word > 0) DGRP_HIP(hipMemcpyAsync(htiles.data(), tiles_base, (size_t)word * 8, hipMemcpyDeviceToHost, stream));
    DGRP_HIP(hipStreamSynchronize(stream));
    for (int64_t r = 0; r < nrec; ++r) {
        int64_t *info = h_info + 4 * r;
        if (h_len[r] == 0) { info[0] = 1; info[1] = 0; info[2] = 0; info[3] = 0; continue; }
        const unsigned long long *g = hg.data() + (size_t)r * 32;
        const int64_t tot = small[(size_t)r] ? (int64_t)g[3] : (int64_t)htiles[(size_t)grand_at[(size_t)r]];
        info[0] = g[0] ? 0 : 1;
        info[1] = tot;
        if (g[2] == 0) { info[2] = tot; info[3] = -tot; }
        else { info[2] = (int64_t)g[1]; info[3] = (int64_t)g[2] - (int64_t)g[1]; }
    }
    return DGRP_OK;
}

// ------------------------------------------------------------------------------------------
// A1  where the records of a FASTA file start (_read_multi_fasta, deepgrp/__main__.py:31-41: a line whose first character is '>'
// opens a record), found on the uploaded file instead of by host passes over it.  A CHUNK starts at byte 0 and at every '>' that
// directly follows a line feed.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fasta_gt_kernel(const uint8_t *__restrict__ raw, int64_t n, int64_t cap,
                                                       unsigned long long *__restrict__ count, int64_t *__restrict__ list)
{
    // 16 bytes per thread (the buffer is a fresh allocation: 16-byte aligned); '>' is rare, so the append is an atomic
    const int64_t nvec = n / 16;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v <= nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p0 = v * 16;
        union { uint4 q; uint8_t b[16]; } u;
        uint8_t *b = u.b;
        int m = 16;
        if (v < nvec) {
            u.q = *(const uint4 *)(raw + p0);
        } else {
            m = (int)(n - p0);
            for (int j = 0; j < m; ++j) b[j] = raw[p0 + j];
        }
        bool any = false;
        for (int j = 0; j < m; ++j) any = any || b[j] == '>';
        if (!any) continue;
        uint8_t prev = p0 > 0 ? raw[p0 - 1] : 0;
        for (int j = 0; j < m; ++j) {
            if (b[j] == '>' && prev == 10 ) {
                const unsigned long long at = atomicAdd(count, 1ull);
                if ((int64_t)at < cap) list[at] = p0 + j;
            }
            prev = b[j];
        }
    }
}

// one thread per chunk: the first line feed at or after its start (n if there is none)
__global__ void __launch_bounds__(256) fasta_firstlf_kernel(const uint8_t *__restrict__ raw, int64_t n, const int64_t *__restrict__ start,
                                                            int64_t nchunks, int64_t *__restrict__ first_lf)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const int64_t stop = c + 1 < nchunks ? start[c + 1] : n;       // a later chunk starts behind a line feed: the search ends before it
    int64_t p = start[c];
    while (p < stop && raw[p] != 10) ++p;
    first_lf[c] = p < stop ? p : n;
}

DGRP_EXPORT int64_t dgrp_fasta_chunks_workspace_bytes(int64_t cap)
{
    if (cap < 0) return 0;
    return 256 + dgrp_align_up((cap + 1) * 8, 256) * 2;
}

// h_start[0] = 0 and the '>' positions in ascending order, h_first_lf[i] as fasta_firstlf_kernel defines it; *n_chunks is the number
// of chunks found -- if it exceeds cap nothing else is valid and the caller repeats the call with cap >= *n_chunks.  Synchronous.
DGRP_EXPORT int dgrp_fasta_chunks(const uint8_t *d_raw, int64_t nbytes, int64_t cap, int64_t *h_start, int64_t *h_first_lf,
                                  int64_t *n_chunks, void *d_work, int64_t work_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    DGRP_REQUIRE(nbytes >= 0 && cap >= 1 && h_start && h_first_lf && n_chunks, "dgrp_fasta_chunks: bad arguments");
    if (nbytes == 0) { *n_chunks = 0; return DGRP_OK; }
    DGRP_REQUIRE(d_raw && d_work, "dgrp_fasta_chunks: NULL pointer");
    DGRP_REQUIRE(((uintptr_t)d_raw & 15) == 0, "dgrp_fasta_chunks: d_raw must be 16-byte aligned");
    if (work_bytes < dgrp_fasta_chunks_workspace_bytes(cap)) {
        dgrp_set_error("dgrp_fasta_chunks: workspace too small");
        return DGRP_ENOMEM;
    }
    unsigned long long *d_count = (unsigned long long *)d_work;
    int64_t *d_list = (int64_t *)((unsigned char *)d_work + 256);
    int64_t *d_lf = (int64_t *)((unsigned char *)d_list + dgrp_align_up((cap + 1) * 8, 256));
    DGRP_HIP(hipMemsetAsync(d_count, 0, 8, stream));
    const int64_t nvec = nbytes / 16 + 1;
    const unsigned grid = (unsigned)(nvec + 255) / 256 > 16384u ? 16384u : (unsigned)((nvec + 255) / 256);
    hipLaunchKernelGGL(fasta_gt_kernel, dim3(grid), dim3(256), 0, stream, d_raw, nbytes, cap - 1, d_count, d_list + 1);
    DGRP_LAUNCH_CHECK();
    unsigned long long found = 0;
    DGRP_HIP(hipMemcpyAsync(&found, d_count, 8, hipMemcpyDeviceToHost, stream));
    DGRP_HIP(hipStreamSynchronize(stream));
    *n_chunks = (int64_t)found + 1;
    if (*n_chunks > cap) return DGRP_OK;
    h_start[0] = 0;
    if (found) {
        DGRP_HIP(hipMemcpyAsync(h_start + 1, d_list + 1, found * 8, hipMemcpyDeviceToHost, stream));
        DGRP_HIP(hipStreamSynchronize(stream));
        std::sort(h_start + 1, h_start + 1 + found);
    }
    DGRP_HIP(hipMemcpyAsync(d_list, h_start, (found + 1) * 8, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(fasta_firstlf_kernel, dim3((unsigned)((found + 1 + 255) / 256)), dim3(256), 0, stream, d_raw, nbytes, d_list,
                       (int64_t)found + 1, d_lf);
    DGRP_LAUNCH_CHECK();
    DGRP_HIP(hipMemcpyAsync(h_first_lf, d_lf, (found + 1) * 8, hipMemcpyDeviceToHost, stream));
    DGRP_HIP(hipStreamSynchronize(stream));
    return DGRP_OK;
}

// ------------------------------------------------------------------------------------------
// A12  the TSV rows of deepgrp/__main__.py:291-292 as text: "<prefix>start\tend\tlabel\n" per row, prefix = "file\theader\t" of the
// row's record (row.contig indexes the prefixes when by_contig, else prefix 0).  Host code: 100 000 rows are a few hundred
// microseconds here and tens of milliseconds as numpy string columns.
// ------------------------------------------------------------------------------------------
static inline char *put_int(char *o, long long v)
{
    char tmp[24];
    int k = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *o++ = '-';
    while (k) *o++ = tmp[--k];
    return o;
}

DGRP_EXPORT int64_t dgrp_format_rows_bound(int64_t nrows, int64_t longest_prefix)
{
    if (nrows < 0 || longest_prefix < 0) return 0;
    return nrows * (longest_prefix + 21 + 21 + 12 + 3) + 1;
}

DGRP_EXPORT int dgrp_format_rows(const char *prefixes, const int64_t *prefix_off, int64_t nprefix, int by_contig,
                                 const dgrp_segment *rows, int64_t nrows, char *out, int64_t cap, int64_t *written)
{
    DGRP_REQUIRE(nrows >= 0 && nprefix >= 1 && prefixes && prefix_off && written && (nrows == 0 || (rows && out)),
                 "dgrp_format_rows: bad arguments");
    int64_t longest = 0;
    for (int64_t i = 0; i < nprefix; ++i) {
        DGRP_REQUIRE(prefix_off[i + 1] >= prefix_off[i], "dgrp_format_rows: prefix offsets must ascend");
        if (prefix_off[i + 1] - prefix_off[i] > longest) longest = prefix_off[i + 1] - prefix_off[i];
    }
    if (cap < dgrp_format_rows_bound(nrows, longest)) {
        dgrp_set_error("dgrp_format_rows: output buffer too small (dgrp_format_rows_bound)");
        return DGRP_ENOMEM;
    }
    char *o = out;
    for (int64_t r = 0; r < nrows; ++r) {
        const int64_t c = by_contig ? rows[r].contig : 0;
        DGRP_REQUIRE(c >= 0 && c < nprefix, "dgrp_format_rows: row %lld names record %lld of %lld", (long long)r, (long long)c, (long long)nprefix);
        const int64_t len = prefix_off[c + 1] - prefix_off[c];
        memcpy(o, prefixes + prefix_off[c], (size_t)len);
        o += len;
        o = put_int(o, rows[r].start); *o++ = '\t';
        o = put_int(o, rows[r].end); *o++ = '\t';
        o = put_int(o, rows[r].label); *o++ = '\n';
    }
    *written = o - out;
    return DGRP_OK;
}

