// sequence.hpp -- input side of the hot path on the host: PNG decode, the dataset iterators and a
// batched feeder that keeps several stereo pairs in flight on the device.  Header-only C++17; needs zlib (-lz) and the C
// ABI of ebvo_hip.h, nothing else -- it replaces the cv::imread calls and the frame loop of the reference, not OpenCV.
//
//   ebvo::read_png_gray           cv::imread(path, cv::IMREAD_GRAYSCALE) for the files the datasets hold: 8-bit
//                                 grayscale PNG (KITTI image_0 / image_1, EuRoC cam0 / cam1); 8-bit grayscale + alpha,
//                                 RGB and RGBA are converted with OpenCV's fixed-point BGR -> gray weights
//                                 ((R * 4899 + G * 9617 + B * 1868 + 8192) >> 14); interlaced, paletted and 16-bit files
//                                 are refused (the iterators print the error and skip the pair, like the reference's
//                                 "Failed to load" paths)
//   ebvo::KittiSequence           KITTIIterator (src/Stereo_Iterator.cpp:84-184): <dir>/image_0|image_1/%06d.png
//   ebvo::EurocSequence           EuRoCIterator (:18-78): one image pair per line of data.csv, <left>/<ts>.png, <right>/<ts>.png
//   ebvo::BatchedStereoFeeder     the frame loop of cmd/main_VO.cpp:99-113 for the resident pipeline: a decoder thread
//                                 reads ahead, `slots` pairs are in flight (ebvo_stereo_upload_slot / _submit / _wait), the
//                                 callback receives every finished pair in sequence order with its results still resident
#ifndef EBVO_SEQUENCE_HPP
#define EBVO_SEQUENCE_HPP

#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <filesystem>
#include <fstream>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../ebvo_hip.h"

namespace ebvo
{

struct GrayImage
{
    int width = 0, height = 0;
    std::vector<uint8_t> pixels; // row-major, tightly packed
    bool empty() const { return pixels.empty(); }
};

namespace detail
{
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
} // namespace detail

// Returns "" on success, otherwise what is wrong with the file.
inline std::string read_png_gray(const std::string &path, GrayImage &out)
{
    out = GrayImage();
    std::ifstream f(path, std::ios::binary);
    if (!f)
        return "cannot open " + path;
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0)
        return path + ": not a PNG file";
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat;
    bool seen_end = false;
    while (pos + 12 <= file.size() && !seen_end)
    {
        const uint32_t len = detail::be32(&file[pos]);
        const char *type = reinterpret_cast<const char *>(&file[pos + 4]);
        if (pos + 12 + (size_t)len > file.size())
            return path + ": truncated chunk";
        const uint8_t *data = &file[pos + 8];
        if (crc32(crc32(0L, Z_NULL, 0), &file[pos + 4], len + 4) != detail::be32(&file[pos + 8 + len]))
            return path + ": chunk checksum mismatch";
        if (!std::memcmp(type, "IHDR", 4))
        {
            if (len != 13)
                return path + ": bad IHDR";
            w = detail::be32(data);
            h = detail::be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        }
        else if (!std::memcmp(type, "IDAT", 4))
            idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4))
            seen_end = true;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0 || w > (1u << 15) || h > (1u << 15))
        return path + ": missing or implausible IHDR";
    if (depth != 8 || interlace != 0 || !(ctype == 0 || ctype == 2 || ctype == 4 || ctype == 6))
        return path + ": only non-interlaced 8-bit gray / gray+alpha / RGB / RGBA PNG files are supported";
    const int ch = ctype == 0 ? 1 : (ctype == 4 ? 2 : (ctype == 2 ? 3 : 4));
    const size_t stride = (size_t)w * ch, raw_size = (stride + 1) * h;
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw_size)
        return path + ": zlib stream does not hold the image";
    // undo the scanline filters in place (PNG specification, section 9)
    std::vector<uint8_t> prev(stride, 0);
    std::vector<uint8_t> pix((size_t)w * h * ch);
    for (uint32_t y = 0; y < h; ++y)
    {
        const uint8_t ft = raw[(stride + 1) * y];
        uint8_t *cur = &raw[(stride + 1) * y + 1];
        for (size_t x = 0; x < stride; ++x)
        {
            const int a = x >= (size_t)ch ? cur[x - ch] : 0, b = prev[x], c = x >= (size_t)ch ? prev[x - ch] : 0;
            int v = cur[x];
            switch (ft)
            {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += detail::paeth(a, b, c); break;
            default: return path + ": unknown scanline filter";
            }
            cur[x] = (uint8_t)v;
        }
        std::memcpy(prev.data(), cur, stride);
        std::memcpy(&pix[(size_t)y * stride], cur, stride);
    }
    out.width = (int)w;
    out.height = (int)h;
    out.pixels.resize((size_t)w * h);
    if (ch <= 2)
        for (size_t k = 0; k < (size_t)w * h; ++k)
            out.pixels[k] = pix[k * ch]; // gray (+ alpha dropped, as cv::imread does)
    else
        for (size_t k = 0; k < (size_t)w * h; ++k)
        {
            const int R = pix[k * ch], G = pix[k * ch + 1], B = pix[k * ch + 2];
            out.pixels[k] = (uint8_t)((R * 4899 + G * 9617 + B * 1868 + 8192) >> 14); // cv::cvtColor BGR2GRAY, 8-bit
        }
    return "";
}

struct StereoImages
{
    GrayImage left, right;
    double timestamp = 0;
    size_t index = 0;
};

class StereoSequence
{
  public:
    virtual ~StereoSequence() = default;
    virtual bool hasNext() = 0;
    virtual bool getNext(StereoImages &frame) = 0; // false: no further pair could be read
    virtual void reset() = 0;
};

// KITTIIterator (src/Stereo_Iterator.cpp:84-184): <dir>/image_0/%06d.png and <dir>/image_1/%06d.png, index = timestamp
class KittiSequence : public StereoSequence
{
  public:
    explicit KittiSequence(const std::string &dataset_path) : dir_(dataset_path)
    {
        std::error_code ec;
        for (const auto &e : std::filesystem::directory_iterator(dir_ + "/image_0/", ec))
            if (e.path().extension() == ".png")
                ++total_;
        if (ec)
            std::fprintf(stderr, "WARNING: Could not scan KITTI image directory: %s/image_0/\n", dir_.c_str());
    }
    bool hasNext() override { return cur_ < total_; }
    void reset() override { cur_ = 0; }
    size_t size() const { return total_; }
    bool getNext(StereoImages &f) override
    {
        if (!hasNext())
            return false;
        char name[32];
        std::snprintf(name, sizeof name, "%06zu.png", cur_);
        const std::string el = read_png_gray(dir_ + "/image_0/" + name, f.left), er = read_png_gray(dir_ + "/image_1/" + name, f.right);
        if (!el.empty() || !er.empty())
        {
            std::fprintf(stderr, "ERROR: Failed to load KITTI images at index %zu (%s%s%s)\n", cur_, el.c_str(),
                         el.empty() || er.empty() ? "" : "; ", er.c_str());
            return false;
        }
        f.timestamp = (double)cur_;
        f.index = cur_++;
        return true;
    }

  private:
    std::string dir_;
    size_t total_ = 0, cur_ = 0;
};

// EuRoCIterator (:18-78): data.csv (first line skipped), "<timestamp>,..." per pair, <left_dir><timestamp>.png
class EurocSequence : public StereoSequence
{
  public:
    EurocSequence(const std::string &csv_path, const std::string &left_dir, const std::string &right_dir)
        : csv_path_(csv_path), left_(left_dir), right_(right_dir)
    {
        csv_.open(csv_path_);
        if (!csv_.is_open())
            std::fprintf(stderr, "ERROR: Could not open: %s\n", csv_path_.c_str());
    }
    bool hasNext() override { return csv_ && csv_.peek() != EOF; }
    void reset() override
    {
        csv_.close();
        csv_.open(csv_path_);
        first_skipped_ = false;
        count_ = 0;
    }
    bool getNext(StereoImages &f) override
    {
        std::string line;
        while (std::getline(csv_, line))
        {
            if (!first_skipped_)
            {
                first_skipped_ = true;
                continue;
            }
            const std::string ts = line.substr(0, line.find(','));
            if (ts.empty())
                continue;
            if (read_png_gray(left_ + ts + ".png", f.left).empty() && read_png_gray(right_ + ts + ".png", f.right).empty())
            {
                f.timestamp = std::stod(ts);
                f.index = count_++;
                return true;
            }
            std::fprintf(stderr, "Skipping image pair: %s\n", ts.c_str());
        }
        return false;
    }

  private:
    std::string csv_path_, left_, right_;
    std::ifstream csv_;
    bool first_skipped_ = false;
    size_t count_ = 0;
};

// Keeps `slots` pairs of a sequence in flight on one context: decode (background thread, `read_ahead` pairs) -> upload ->
// submit -> wait -> callback.  The callback runs on the calling thread, in sequence order, while the results of that pair
// are still resident in its slot (fetch them with ebvo_stereo_fetch_slot / _fetch_begin, or run ebvo_stereo_finalize).
class BatchedStereoFeeder
{
  public:
    using Callback = std::function<void(const StereoImages &frame, int slot, const ebvo_stereo_counts &counts)>;

    BatchedStereoFeeder(ebvo_ctx *ctx, int slots, int read_ahead = 4) : ctx_(ctx), slots_(slots < 1 ? 1 : slots), ahead_(read_ahead)
    {
        status_ = ebvo_stereo_set_slots(ctx_, slots_);
    }
    int status() const { return status_; }
    size_t skipped() const { return skipped_; } // pairs whose two images differ in size (reported, not processed)

    // runs the whole sequence (or max_frames of it); returns the number of pairs processed
    size_t run(StereoSequence &seq, const ebvo_stereo_params &params, const Callback &done, size_t max_frames = (size_t)-1)
    {
        if (status_ != EBVO_OK)
            return 0;
        Reader reader(seq, ahead_, max_frames);
        std::vector<StereoImages> in_slot((size_t)slots_);
        size_t submitted = 0, completed = 0;
        for (int k = 0; k < slots_; ++k)
            if (!launch(reader, params, k, in_slot))
                break;
            else
                ++submitted;
        while (completed < submitted)
        {
            const int k = (int)(completed % (size_t)slots_);
            ebvo_stereo_counts c;
            const int rc = ebvo_stereo_wait(ctx_, k, &c);
            if (rc != EBVO_OK)
            {
                std::fprintf(stderr, "[ebvo] wait: %s (%s)\n", ebvo_strerror(rc), ebvo_last_error(ctx_));
                status_ = rc;
                break;
            }
            done(in_slot[(size_t)k], k, c);
            ++completed;
            if (status_ == EBVO_OK && launch(reader, params, k, in_slot))
                ++submitted;
        }
        return completed;
    }

    // The same loop with the stages after the first NCC pass as well (ebvo_stereo_finalize_submit / _wait): when the counts
    // of a pair have arrived its chain is enqueued at once, without a host synchronisation between its stages, and the pair
    // stays in its slot until the chain has run; up to `chains` chains are in flight next to the pairs still in their first
    // stage, and a slot receives its next pair when its chain has been retired.  `done` runs on the calling thread, in
    // sequence order, with the final pairs of the frame still resident in its slot (ebvo_stereo_fetch_final, or
    // ebvo_temporal_* against a keyframe).
    using ChainCallback = std::function<void(const StereoImages &frame, int slot, const ebvo_stereo_counts &counts,
                                             const ebvo_finalize_counts &final_counts)>;
    size_t run_chain(StereoSequence &seq, const ebvo_stereo_params &params, const ebvo_finalize_params &fin,
                     const ebvo_stereo_calib *calib, const ChainCallback &done, int chains = 2, size_t max_frames = (size_t)-1)
    {
        if (status_ != EBVO_OK)
            return 0;
        if (chains < 1)
            chains = 1;
        if (chains > slots_)
            chains = slots_;
        Reader reader(seq, ahead_, max_frames);
        std::vector<StereoImages> in_slot((size_t)slots_);
        std::deque<std::pair<int, ebvo_stereo_counts>> in_chain; // slots whose chain is enqueued, oldest first
        std::deque<int> stage1;                                  // slots whose first stage is submitted, oldest first
        size_t finished = 0;
        auto retire = [&]() {
            const int k = in_chain.front().first;
            const ebvo_stereo_counts c = in_chain.front().second;
            in_chain.pop_front();
            ebvo_finalize_counts fc;
            const int rc = ebvo_stereo_finalize_wait(ctx_, k, &fc);
            if (rc != EBVO_OK)
            {
                std::fprintf(stderr, "[ebvo] finalize_wait: %s (%s)\n", ebvo_strerror(rc), ebvo_last_error(ctx_));
                status_ = rc;
                return;
            }
            done(in_slot[(size_t)k], k, c, fc);
            ++finished;
            if (launch(reader, params, k, in_slot)) // the slot is free again
                stage1.push_back(k);
        };
        for (int k = 0; k < slots_; ++k)
            if (launch(reader, params, k, in_slot))
                stage1.push_back(k);
            else
                break;
        while (status_ == EBVO_OK && (!stage1.empty() || !in_chain.empty()))
        {
            if (!stage1.empty())
            {
                const int k = stage1.front();
                stage1.pop_front();
                ebvo_stereo_counts c;
                int rc = ebvo_stereo_wait(ctx_, k, &c);
                if (rc == EBVO_OK)
                    rc = ebvo_stereo_finalize_submit(ctx_, k, &fin, calib);
                if (rc != EBVO_OK)
                {
                    std::fprintf(stderr, "[ebvo] pair %zu: %s (%s)\n", in_slot[(size_t)k].index, ebvo_strerror(rc), ebvo_last_error(ctx_));
                    status_ = rc;
                    break;
                }
                in_chain.emplace_back(k, c);
            }
            while (status_ == EBVO_OK && !in_chain.empty() && ((int)in_chain.size() >= chains || stage1.empty()))
                retire();
        }
        if (status_ != EBVO_OK) // leave nothing in flight behind an error
        {
            ebvo_finalize_counts fc;
            ebvo_stereo_counts c;
            for (auto &pr : in_chain)
                (void)ebvo_stereo_finalize_wait(ctx_, pr.first, &fc);
            for (int k : stage1)
                (void)ebvo_stereo_wait(ctx_, k, &c);
        }
        return finished;
    }

  private:
    // decode on a background thread, `ahead` pairs in front of the consumer
    class Reader
    {
      public:
        Reader(StereoSequence &seq, int ahead, size_t max_frames)
            : decoder_([this, &seq, ahead, max_frames] {
                  size_t produced = 0;
                  while (produced < max_frames && !stop_.load())
                  {
                      StereoImages f;
                      if (!seq.getNext(f))
                          break;
                      std::unique_lock<std::mutex> lk(m_);
                      cv_space_.wait(lk, [&] { return (int)ready_.size() < ahead || stop_.load(); });
                      if (stop_.load())
                          break;
                      ready_.push_back(std::move(f));
                      ++produced;
                      cv_item_.notify_one();
                  }
                  std::lock_guard<std::mutex> lk(m_);
                  eof_ = true;
                  cv_item_.notify_one();
              })
        {
        }
        ~Reader()
        {
            {
                std::lock_guard<std::mutex> lk(m_);
                stop_.store(true); // unblock a decoder that still waits for space
                ready_.clear();
                cv_space_.notify_all();
            }
            decoder_.join();
        }
        bool take(StereoImages &f)
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_item_.wait(lk, [&] { return !ready_.empty() || eof_; });
            if (ready_.empty())
                return false;
            f = std::move(ready_.front());
            ready_.pop_front();
            cv_space_.notify_one();
            return true;
        }

      private:
        std::deque<StereoImages> ready_;
        std::mutex m_;
        std::condition_variable cv_space_, cv_item_;
        bool eof_ = false;
        std::atomic<bool> stop_{false};
        std::thread decoder_; // last member: starts when everything above exists
    };

    // next pair of the sequence into slot k: upload + submit; false = the sequence has ended or a call failed (status_)
    bool launch(Reader &reader, const ebvo_stereo_params &params, int k, std::vector<StereoImages> &in_slot)
    {
        StereoImages f;
        // left and right must share one size (the TOED object is built once from the left image); a pair
        // that does not is reported and skipped, as the reference does for frames that fail to load
        for (;;)
        {
            if (!reader.take(f))
                return false;
            if (f.left.width == f.right.width && f.left.height == f.right.height &&
                f.left.pixels.size() == (size_t)f.left.width * f.left.height &&
                f.right.pixels.size() == (size_t)f.right.width * f.right.height)
                break;
            std::fprintf(stderr, "[ebvo] pair %zu skipped: left %dx%d, right %dx%d\n", f.index, f.left.width, f.left.height,
                         f.right.width, f.right.height);
            ++skipped_;
        }
        int rc = ebvo_stereo_upload_slot(ctx_, k, f.left.pixels.data(), f.right.pixels.data(), f.left.height, f.left.width,
                                         f.left.width, f.right.width);
        if (rc == EBVO_OK)
            rc = ebvo_stereo_submit(ctx_, k, &params);
        if (rc != EBVO_OK)
        {
            std::fprintf(stderr, "[ebvo] pair %zu: %s (%s)\n", f.index, ebvo_strerror(rc), ebvo_last_error(ctx_));
            status_ = rc;
            return false;
        }
        in_slot[(size_t)k] = std::move(f);
        return true;
    }

  private:
    ebvo_ctx *ctx_;
    int slots_, ahead_, status_ = EBVO_OK;
    size_t skipped_ = 0;
};

} // namespace ebvo
#endif // EBVO_SEQUENCE_HPP
