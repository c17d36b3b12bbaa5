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
        window_ = std::max(4u, 2 * threads);
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

private:
    struct Result {
        std::string out;
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
            }
            cv_done_.notify_all();
        }
    }

    // `guess`: distance to the next candidate, i.e. the compressed size if both are real boundaries
    // capped: the limits that keep a SPECULATIVE candidate from running through the rest of the file; a member the chain
    // has actually arrived at is real, however large, and is inflated without them
    void inflate_member(size_t p, size_t guess, Result &r, bool capped = true) const
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        r.state = 3;
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return;
        zs.next_in = const_cast<Bytef *>(map_ + p);
        // a real member is at most kMaxMember compressed bytes long (open() checked the candidate gaps); a false
        // candidate that happens to parse as deflate data must not run through the rest of the file, and no member may
        // grow its output without bound
        size_t avail = capped ? std::min(size_ - p, kMaxMember + (1u << 16)) : size_ - p;
        const size_t max_output = capped ? kMaxOutput : SIZE_MAX / 2;
        r.out.resize(std::max<size_t>(1u << 16, 4 * guess));
        size_t produced = 0;
        for (;;) {
            const uInt in_chunk = (uInt)std::min<size_t>(avail, 1u << 30);
            zs.avail_in = in_chunk;
            if (produced == r.out.size()) {
                if (produced >= max_output) break;
                r.out.resize(std::min(r.out.size() * 2, max_output));
            }
            const uInt out_chunk = (uInt)std::min<size_t>(r.out.size() - produced, 1u << 30);
            zs.next_out = (Bytef *)&r.out[produced];
            zs.avail_out = out_chunk;
            const int rc = inflate(&zs, Z_NO_FLUSH);
            produced += out_chunk - zs.avail_out;
            avail -= in_chunk - zs.avail_in;
            if (rc == Z_STREAM_END) {
                r.out.resize(produced);
                r.end = (size_t)(zs.next_in - map_);
                r.state = 2;
                break;
            }
            if (rc != Z_OK && rc != Z_BUF_ERROR) break;              // not a deflate stream: a false candidate
            if (zs.avail_in == 0 && avail == 0 && rc != Z_STREAM_END && zs.avail_out != 0) break; // truncated, or longer than a member may be
        }
        inflateEnd(&zs);
        if (r.state != 2) r.out.clear();
    }

    // move to the next member of the chain; false at the end of the file
    bool advance()
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (cur_ != SIZE_MAX) {
            std::string().swap(res_[cur_].out);
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
        for (size_t j = chain_idx_; j < i; ++j) std::string().swap(res_[j].out);   // candidates the chain jumped over
        chain_idx_ = i;
        cv_work_.notify_all();
        cv_done_.wait(lk, [&] { return res_[i].state >= 2; });
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
    size_t next_task_ = 0, chain_idx_ = 0, chain_off_ = 0, window_ = 8;
    size_t cur_ = SIZE_MAX, cur_pos_ = 0;
    bool stop_ = false, eof_ = false;
};

} // namespace fastx
