// gzmembers.h -- parallel inflate of multi-member gzip files (bgzip output, `cat a.fastq.gz b.fastq.gz > all.fastq.gz`,
// sequencer output concatenated per run).  One deflate stream can only be inflated front to back, but the members
// of such a file are independent streams.  Their boundaries are not indexed anywhere, so they are found speculatively:
// every offset that looks like a member header (1f 8b 08, reserved flag bits clear) is a candidate and is inflated by
// a worker; the consumer walks the chain -- member at offset 0, then the member that starts exactly where the previous
// one ended, ... -- and hands out their bytes in order.  A candidate that is not a real boundary lies inside a member,
// is never reached by the chain, and its (almost always immediately failing) inflate is discarded.
// Files with a single member, or with members too large to hold a few of them in memory, are left to the sequential
// zlib reader.
#pragma once

#include "pgz.h"      // the deflate decoder, the CRC-32 by carry-less multiplication

#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace fastx {

class GzMembers {
public:
    ~GzMembers()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        cv_done_.notify_all();
        for (auto &t : th_) t.join();
        if (map_) munmap(const_cast<unsigned char *>(map_), size_);
    }

    // true: the file is a gzip file with several members of manageable size; inflating starts on `threads` threads
    bool open(const std::string &path, unsigned threads)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) { ::close(fd); return false; }
        size_ = (size_t)sb.st_size;
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return false;
        map_ = (const unsigned char *)m;
        if (!header_at(0)) return false;
        for (size_t p = 0; p + 10 <= size_;) {                      // candidate member headers
            const unsigned char *q = (const unsigned char *)memchr(map_ + p, 0x1f, size_ - p - 9);
            if (!q) break;
            p = (size_t)(q - map_);
            if (header_at(p)) cand_.push_back(p);
            ++p;
        }
        if (cand_.size() < 4) return false;
        size_t max_gap = 0;
        for (size_t i = 0; i < cand_.size(); ++i)
            max_gap = std::max(max_gap, (i + 1 < cand_.size() ? cand_[i + 1] : size_) - cand_[i]);
        if (max_gap > kMaxMember) return false;                   // members that large: a few of them would not fit in memory
        res_.resize(cand_.size());
        // Candidates are only byte patterns (about one false `1f 8b 08` per 2^27 bytes of deflate data), so a large
        // single-member file can pass the tests above.  Commit to parallel mode only if the first member really ends
        // on another candidate (or at the end of the file) within the size limit; its output is kept, not redone.
        {
            Result r0;
            inflate_member(0, cand_[1], r0);
            const bool chained = r0.state == 2 && (r0.end >= size_ || std::binary_search(cand_.begin(), cand_.end(), r0.end) ||
                                                   only_zeros_from(r0.end));
            if (!chained) return false;
            res_[0].out.swap(r0.out);
            res_[0].end = r0.end;
            res_[0].state = 2;
            next_task_ = 1;
        }
        // (candidates in flight ahead of the chain.  bgzip's members are 64 KB: a window of two per thread had the consumer and the
        // workers hand every member over one by one -- a lock, a wake-up of all workers and a wake-up of the consumer per member,
        // 13 us each, which was the reader's rate)
        {   // ... as many as ~1 GB of compressed members (their output is a few times that), never fewer than two per thread
            const size_t avg = std::max<size_t>(1, size_ / std::max<size_t>(1, cand_.size()));
            window_ = std::max<size_t>(std::max(4u, 2 * threads), std::min<size_t>(std::max(64u, 8 * threads), (1u << 30) / avg));
        }
        for (unsigned t = 0; t < std::max(1u, threads); ++t) th_.emplace_back([this] { worker(); });
        return true;
    }

    // next bytes of the decompressed stream; 0 at the end
    size_t read(char *dst, size_t n)
    {
        size_t got = 0;
        while (got < n && !eof_) {
            if (cur_ == SIZE_MAX || cur_pos_ == res_[cur_].out.size()) {
                if (!advance()) break;
                continue;
            }
            const size_t take = std::min(n - got, res_[cur_].out.size() - cur_pos_);
            memcpy(dst + got, res_[cur_].out.data() + cur_pos_, take);
            cur_pos_ += take;
            got += take;
        }
        return got;
    }

    // the next member's bytes as a whole (the buffer changes hands, nothing is copied); false at the end.  read() may follow -- it goes
    // on with the member after the one taken -- but not the other way round in the middle of a member.
    bool take(std::vector<char> &dst)
    {
        if (eof_ || !advance()) return false;
        dst.swap(res_[cur_].out);
        res_[cur_].out.clear();
        cur_pos_ = 0;
        return true;
    }

private:
    struct Result {
        std::vector<char> out;
        size_t end = 0;
        int state = 0;   // 0 not started, 1 running, 2 inflated, 3 not a member / corrupt
    };

    bool only_zeros_from(size_t p) const
    {
        for (; p < size_; ++p)
            if (map_[p] != 0) return false;
        return true;
    }

    bool header_at(size_t p) const
    {
        return p + 10 <= size_ && map_[p] == 0x1f && map_[p + 1] == 0x8b && map_[p + 2] == 8 && (map_[p + 3] & 0xE0) == 0;
    }

    void worker()
    {
        for (;;) {
            size_t i;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || (next_task_ < cand_.size() && next_task_ < chain_idx_ + window_); });
                if (stop_) return;
                i = next_task_++;
                if (cand_[i] < chain_off_) {                        // the chain is past it: it was inside a member
                    res_[i].state = 3;
                    continue;
                }
                res_[i].state = 1;
            }
            Result r;
            inflate_member(cand_[i], (i + 1 < cand_.size() ? cand_[i + 1] : size_) - cand_[i], r);
            {
                std::lock_guard<std::mutex> lk(mu_);
                res_[i].out.swap(r.out);
                res_[i].end = r.end;
                res_[i].state = r.state;
                if (waiting_for_ != i) continue;               // (the consumer is woken for the member it waits for, not for every one)
            }
            cv_done_.notify_all();
        }
    }

    // `guess`: distance to the next candidate, i.e. the compressed size if both are real boundaries
    // capped: the limits that keep a SPECULATIVE candidate from running through the rest of the file; a member the chain
    // has actually arrived at is real, however large, and is inflated without them
    void inflate_member(size_t p, size_t guess, Result &r, bool capped = true) const
    {
        // (round 4: the member goes through this repository's own deflate decoder -- pgz.h: 2.2 ns per byte against zlib's 3-4.5 --
        // into 16-bit symbols; a member has nothing before it, so a symbol that is no byte is a reference behind its start: invalid.
        // CRC-32 and length are checked against the member's trailer like zlib does.)
        using namespace pgz_detail;
        (void)guess;
        r.state = 3;
        // a real member is at most kMaxMember compressed bytes long (open() checked the candidate gaps); a false
        // candidate that happens to parse as deflate data must not run through the rest of the file, and no member may
        // grow its output without bound
        const size_t avail = capped ? std::min(size_ - p, kMaxMember + (1u << 16)) : size_ - p;
        const size_t max_output = capped ? kMaxOutput : SIZE_MAX / 8;
        const size_t h = gzip_header_len(map_ + p, avail);
        if (!h) return;
        static thread_local ChunkOut co;
        const uint8_t *base = map_ + p, *end = base + avail;
        const bool ok = decode_from(base, end, (uint64_t)h * 8, ~0ull, co, WIN + max_output, nullptr, 1 << 30, true) && co.final_block;
        if (ok) {
            const size_t n = co.n - WIN, t = (size_t)((co.end_bit + 7) / 8);         // the trailer behind the final block's last byte
            if (t + 8 <= size_ - p) {
                r.out.resize(n);
                const uint16_t *sy = co.sym.data() + WIN;
                uint8_t *o = reinterpret_cast<uint8_t *>(&r.out[0]);
                uint32_t any = 0;
                size_t i = 0;
#if defined(__SSE2__)
                __m128i acc = _mm_setzero_si128();
                for (; i + 16 <= n; i += 16) {
                    const __m128i x = _mm_loadu_si128((const __m128i *)(sy + i)), y = _mm_loadu_si128((const __m128i *)(sy + i + 8));
                    acc = _mm_or_si128(acc, _mm_or_si128(x, y));
                    _mm_storeu_si128((__m128i *)(o + i), _mm_packus_epi16(x, y));
                }
                acc = _mm_srli_epi16(acc, 8);
                any = (uint32_t)(_mm_movemask_epi8(_mm_cmpeq_epi8(acc, _mm_setzero_si128())) != 0xFFFF);
#endif
                for (; i < n; ++i) {
                    any |= (uint32_t)(sy[i] >> 8);
                    o[i] = (uint8_t)sy[i];
                }
                uint32_t crc, isize;
                memcpy(&crc, base + t, 4);
                memcpy(&isize, base + t + 4, 4);
                if (!any && isize == (uint32_t)n && crc == crc32_bytes((uint32_t)crc32(0L, Z_NULL, 0), o, n)) {
                    r.end = p + t + 8;
                    r.state = 2;
                }
            }
        }
        if (co.sym.size() > (64u << 20)) co.sym.release();      // (a thread keeps its symbol buffer from member to member -- unless a member was huge)
        if (r.state != 2) r.out.clear();
    }

    // move to the next member of the chain; false at the end of the file
    bool advance()
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (cur_ != SIZE_MAX) {
            std::vector<char>().swap(res_[cur_].out);
            cur_pos_ = 0;
            chain_off_ = res_[cur_].end;
        }
        if (chain_off_ >= size_) { eof_ = true; return false; }
        // trailing zero padding is legal after the last member
        const auto it = std::lower_bound(cand_.begin(), cand_.end(), chain_off_);
        if (it == cand_.end() || *it != chain_off_) {
            for (size_t p = chain_off_; p < size_; ++p)
                if (map_[p] != 0) throw std::runtime_error("gzip file: garbage after a member");
            eof_ = true;
            return false;
        }
        const size_t i = (size_t)(it - cand_.begin());
        for (size_t j = chain_idx_; j < i; ++j) std::vector<char>().swap(res_[j].out);   // candidates the chain jumped over
        const bool workers_wait = next_task_ >= chain_idx_ + window_;      // (they wait only with the window exhausted)
        chain_idx_ = i;
        if (workers_wait) cv_work_.notify_all();
        if (res_[i].state < 2) {
            waiting_for_ = i;
            cv_done_.wait(lk, [&] { return res_[i].state >= 2; });
            waiting_for_ = SIZE_MAX;
        }
        if (res_[i].state != 2) {
            // The chain has arrived here from a verified member's end, so this IS a member.  The speculative inflate may
            // have given up only because of its size caps (a genuine member beyond 256 MB compressed that also contains a
            // false header candidate passes open()'s gap test): inflate it again without them before calling it corrupt.
            lk.unlock();
            Result r;
            inflate_member(cand_[i], (i + 1 < cand_.size() ? cand_[i + 1] : size_) - cand_[i], r, false);
            lk.lock();
            if (r.state != 2) throw std::runtime_error("gzip file: corrupt member");
            res_[i].out.swap(r.out);
            res_[i].end = r.end;
            res_[i].state = 2;
        }
        cur_ = i;
        cur_pos_ = 0;
        return true;
    }

    static constexpr size_t kMaxMember = 256u << 20;      // compressed bytes of one member in parallel mode
    static constexpr size_t kMaxOutput = 4ull << 30;      // inflated bytes of one member

    const unsigned char *map_ = nullptr;
    size_t size_ = 0;
    std::vector<size_t> cand_;
    std::vector<Result> res_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    size_t next_task_ = 0, chain_idx_ = 0, chain_off_ = 0, window_ = 8, waiting_for_ = SIZE_MAX;
    size_t cur_ = SIZE_MAX, cur_pos_ = 0;
    bool stop_ = false, eof_ = false;
};

} // namespace fastx
