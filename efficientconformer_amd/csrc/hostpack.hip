// Host side of the batching front door: the reference's collate (utils/preprocessing.py:33-45: sort by length, zero-pad to the longest
// utterance) as a multi-threaded copy of the utterances' samples into ONE staging buffer (pinned, so that the H2D copy that follows is a
// single DMA).  In Python that copy is the front door's bottleneck: a 256-utterance LibriSpeech batch is 190 MB against 5.6 ms of GPU
// work, and per-row interpreter overhead + the GIL hold 8 numpy threads at ~7 GB/s (tools/front_door_profile.py).  No device code here.
#include "../../include/effconf.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

int ec_fail(const char* msg);

extern "C" int effconf_host_pack_rows(const float* const* src, const int64_t* len, int32_t n, float* dst, int64_t pitch, int32_t zero_pad,
                                      int32_t threads) {
    if (n == 0) return 0;
    if (!src || !len || !dst || n < 0 || pitch < 0) return ec_fail("effconf_host_pack_rows: null / negative argument");
    for (int i = 0; i < n; ++i)
        if (len[i] < 0 || len[i] > pitch || (len[i] > 0 && !src[i])) return ec_fail("effconf_host_pack_rows: row longer than the pitch or null row");
    auto copy_rows = [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i) {
            float* d = dst + (size_t)i * pitch;
            if (len[i] > 0) std::memcpy(d, src[i], (size_t)len[i] * sizeof(float));
            if (zero_pad && len[i] < pitch) std::memset(d + len[i], 0, (size_t)(pitch - len[i]) * sizeof(float));
        }
    };
    int nt = std::max(1, std::min<int>(threads, n));
    if (nt == 1) { copy_rows(0, n); return 0; }
    // contiguous row chunks of about equal byte counts (rows arrive sorted by length: equal ROW counts would leave the first thread
    // with several times the last one's bytes)
    double total = 0;
    for (int i = 0; i < n; ++i) total += (double)(zero_pad ? pitch : len[i]) + 64.0;
    std::vector<int> cut(1, 0);
    double acc = 0;
    for (int i = 0; i < n; ++i) {
        acc += (double)(zero_pad ? pitch : len[i]) + 64.0;
        if (acc >= total * (double)cut.size() / nt && (int)cut.size() < nt) cut.push_back(i + 1);
    }
    cut.push_back(n);
    std::vector<std::thread> pool;
    pool.reserve(cut.size());
    for (size_t t = 1; t + 1 < cut.size(); ++t)
        if (cut[t + 1] > cut[t]) pool.emplace_back(copy_rows, cut[t], cut[t + 1]);
    copy_rows(cut[0], cut[1]);
    for (auto& th : pool) th.join();
    return 0;
}
