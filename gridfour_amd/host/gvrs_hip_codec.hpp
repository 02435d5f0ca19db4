// gvrs_hip_codec.hpp -- C++ host-side mirror of Gridfour's compression plug-in interface for the
// MI355X codec (header only; links against libgvrs_hip.so through include/gvrs_hip_codec.h).
//
// The reference is Java; there is no JDK in the build image, so the host side above the C ABI is
// written in C++ with the reference's own names, argument meaning and error behaviour:
//   org.gridfour.compress.ICompressionEncoder   core/src/main/java/org/gridfour/compress/ICompressionEncoder.java:46-92
//   org.gridfour.compress.ICompressionDecoder   .../ICompressionDecoder.java:49-107
//   org.gridfour.compress.CodecHuffman          .../CodecHuffman.java:50-260
// Java `null` results are empty std::optional, Java IOException is gridfour::IOException.
// The Java adapter that binds the same C ABI through JNI is in gridfour_amd/java/.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gvrs_hip_codec.h"

namespace gridfour {

struct IOException : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct ArrayIndexOutOfBoundsException : std::out_of_range {
    using std::out_of_range::out_of_range;
};

/** An interface defining a coder for a Gridfour raster data compression implementation. */
class ICompressionEncoder {
public:
    virtual ~ICompressionEncoder() = default;
    /** Encodes the tile (row-major, nRows*nCols values); empty optional == Java null. */
    virtual std::optional<std::vector<uint8_t>> encode(int codecIndex, int nRows, int nCols, const std::vector<int32_t> &values) = 0;
    virtual std::optional<std::vector<uint8_t>> encodeFloats(int codecIndex, int nRows, int nCols, const std::vector<float> &values) = 0;
    virtual bool implementsFloatingPointEncoding() const = 0;
    virtual bool implementsIntegerEncoding() const = 0;
};

/** An interface defining a decoder for a Gridfour raster data compression implementation. */
class ICompressionDecoder {
public:
    virtual ~ICompressionDecoder() = default;
    /** Decodes the packing; throws IOException on an incompatible packing. */
    virtual std::vector<int32_t> decode(int nRows, int nColumns, const std::vector<uint8_t> &packing) = 0;
    virtual std::optional<std::vector<float>> decodeFloats(int nRows, int nColumns, const std::vector<uint8_t> &packing) = 0;
    virtual void analyze(int nRows, int nColumns, const std::vector<uint8_t> &packing) = 0;
    virtual void reportAnalysisData(std::FILE *ps, int nTilesInRaster) = 0;
    virtual void clearAnalysisData() = 0;
};

/** Drop-in for org.gridfour.compress.CodecHuffman, computed on the GPU (no CPU fallback). */
class CodecHuffmanHip : public ICompressionEncoder, public ICompressionDecoder {
public:
    explicit CodecHuffmanHip(int device = 0)
    {
        const gf_status s = gf_context_create(device, &ctx_);
        if (s != GF_OK) throw std::runtime_error(std::string("CodecHuffmanHip: ") + gf_status_string(s) + " [" + gf_last_error() + "]");
    }
    ~CodecHuffmanHip() override { gf_context_destroy(ctx_); }
    CodecHuffmanHip(const CodecHuffmanHip &) = delete;
    CodecHuffmanHip &operator=(const CodecHuffmanHip &) = delete;

    std::optional<std::vector<uint8_t>> encode(int codecIndex, int nRows, int nCols, const std::vector<int32_t> &values) override
    {
        if ((size_t)nRows * (size_t)nCols != values.size()) throw std::invalid_argument("values.length != nRows*nCols");
        std::vector<uint8_t> out(gf_huffman_max_packing(nRows, nCols));
        size_t n = 0;
        const gf_status s = gf_huffman_encode_i32(ctx_, codecIndex, nRows, nCols, values.data(), out.data(), out.size(), &n);
        if (s == GF_DECLINED) return std::nullopt;                       // CodecHuffman.java:80-82
        if (s == GF_ERR_BOUNDS) throw ArrayIndexOutOfBoundsException("PredictorModelLinear needs nCols >= 2");
        check(s, "gf_huffman_encode_i32");
        out.resize(n);
        return out;
    }
    std::optional<std::vector<uint8_t>> encodeFloats(int, int, int, const std::vector<float> &) override { return std::nullopt; }
    bool implementsFloatingPointEncoding() const override { return false; }
    bool implementsIntegerEncoding() const override { return true; }

    std::vector<int32_t> decode(int nRows, int nColumns, const std::vector<uint8_t> &packing) override
    {
        std::vector<int32_t> out((size_t)nRows * (size_t)nColumns);
        const gf_status s = gf_huffman_decode_i32(ctx_, nRows, nColumns, packing.data(), packing.size(), out.data());
        if (s == GF_ERR_FORMAT || s == GF_ERR_BOUNDS) throw IOException(gf_status_string(s));   // CodecHuffman.java:155-169
        check(s, "gf_huffman_decode_i32");
        return out;
    }
    std::optional<std::vector<float>> decodeFloats(int, int, const std::vector<uint8_t> &) override { return std::nullopt; }

    // statistics (CodecHuffman.java:172-234 over CodecStats.java): the packing is Huffman-decoded and its M32 bytes are
    // histogrammed on the GPU (gf_huffman_analyze_batch); the sums live here
    void analyze(int nRows, int nColumns, const std::vector<uint8_t> &packing) override
    {
        const uint64_t offsets[2] = {0, packing.size()};
        std::vector<uint8_t> padded(packing);
        padded.resize(packing.size() + 16);
        int32_t st = 0;
        check(gf_huffman_analyze_batch(ctx_, nRows, nColumns, 1, padded.data(), offsets, stats_, &st), "gf_huffman_analyze_batch");
        if (st != GF_OK) throw IOException(gf_status_string((gf_status)st));
    }
    // analyze() of a whole batch in one GPU pass; returns the per-packing status
    std::vector<int32_t> analyzeBatch(int nRows, int nColumns, size_t nTiles, const uint8_t *blob, const uint64_t *offsets)
    {
        std::vector<int32_t> st(nTiles);
        check(gf_huffman_analyze_batch(ctx_, nRows, nColumns, nTiles, blob, offsets, stats_, st.data()), "gf_huffman_analyze_batch");
        return st;
    }
    const gf_codec_stats *analysisData() const { return stats_; }
    void reportAnalysisData(std::FILE *ps, int nTilesInRaster) override
    {
        std::fprintf(ps, "Gridfour_Huffman                               Compressed Output    |       Predictor Residuals\n");
        if (stats_[5].n_tiles == 0 || nTilesInRaster == 0) {
            std::fprintf(ps, "   Tiles Compressed:  0\n");
            return;
        }
        std::fprintf(ps, "  Predictor                Times Used        bits/sym    bits/tile  |  m32 avg-len   avg-unique  entropy | bits in tree\n");
        static const char *names[6] = {"None", "Differencing", "Linear", "Triangle", "DifferencingWithNulls", "All Predictors"};
        for (int p = 1; p < 6; p++) {
            const gf_codec_stats &r = stats_[p];
            const double n = (double)r.n_tiles, nm = (double)r.n_m32_counted;
            std::fprintf(ps, "   %-20.20s %8ld (%4.1f %%)     %5.2f  %12.1f   | %10.1f      %6.1f    %6.2f   | %6.1f\n", names[p],
                         (long)r.n_tiles, 100.0 * n / nTilesInRaster, r.n_symbols ? 8.0 * r.n_bytes / r.n_symbols : 0.0,
                         n ? r.n_bytes / n * 8 : 0.0, nm ? r.sum_length_m32 / nm : 0.0, n ? r.sum_observed_m32 / n : 0.0,
                         nm ? r.sum_entropy_m32 / nm : 0.0, n ? r.n_bits_overhead / n : 0.0);
        }
    }
    void clearAnalysisData() override { std::memset(stats_, 0, sizeof stats_); }

    // ---- batched forms: what the GPU is for (one launch per batch, not per tile) ----
    struct Batch {
        std::vector<uint8_t> blob;          // packings, concatenated in tile order
        std::vector<uint64_t> offsets;      // nTiles + 1
        std::vector<uint8_t> predictors;    // PredictorModelType code per tile
        std::vector<int32_t> status;        // gf_status per tile (GF_DECLINED == Java null)
    };
    Batch encodeBatch(int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values)
    {
        Batch b;
        b.offsets.resize(nTiles + 1);
        b.predictors.resize(nTiles);
        b.status.resize(nTiles);
        b.blob.resize(nTiles * gf_huffman_default_stride(nRows, nCols));
        gf_status s = gf_huffman_encode_batch_i32(ctx_, codecIndex, nRows, nCols, nTiles, values, b.blob.data(), b.blob.size(),
                                                  b.offsets.data(), b.predictors.data(), b.status.data());
        if (s == GF_ERR_CAPACITY) {
            b.blob.resize(b.offsets[nTiles]);
            s = gf_huffman_encode_batch_i32(ctx_, codecIndex, nRows, nCols, nTiles, values, b.blob.data(), b.blob.size(),
                                            b.offsets.data(), b.predictors.data(), b.status.data());
        }
        check(s, "gf_huffman_encode_batch_i32");
        b.blob.resize(b.offsets[nTiles]);
        return b;
    }
    std::vector<int32_t> decodeBatch(int nRows, int nCols, const Batch &b, std::vector<int32_t> *status = nullptr)
    {
        const size_t nTiles = b.offsets.size() - 1;
        std::vector<int32_t> out(nTiles * (size_t)nRows * (size_t)nCols), st(nTiles);
        std::vector<uint8_t> padded(b.blob);
        padded.resize(padded.size() + 16);
        check(gf_huffman_decode_batch_i32(ctx_, nRows, nCols, nTiles, padded.data(), b.offsets.data(), out.data(), st.data()),
              "gf_huffman_decode_batch_i32");
        if (status) *status = st;
        return out;
    }
    gf_context *context() { return ctx_; }

private:
    static void check(gf_status s, const char *where)
    {
        if (s < 0) throw std::runtime_error(std::string(where) + ": " + gf_status_string(s) + " [" + gf_last_error() + "]");
    }
    gf_context *ctx_ = nullptr;
    gf_codec_stats stats_[6] = {};          // by predictor code 0..4, [5] = all predictors
};

/** Drop-in for org.gridfour.compress.CodecFloat (CodecFloat.java:328-458): byte planes on the GPU,
 *  Deflate of the five planes on the host's zlib. */
class CodecFloatHip : public ICompressionEncoder, public ICompressionDecoder {
public:
    explicit CodecFloatHip(int device = 0, int zlibLevel = 9) : level_(zlibLevel)
    {
        const gf_status s = gf_context_create(device, &ctx_);
        if (s != GF_OK) throw std::runtime_error(std::string("CodecFloatHip: ") + gf_status_string(s) + " [" + gf_last_error() + "]");
    }
    ~CodecFloatHip() override { gf_context_destroy(ctx_); }
    CodecFloatHip(const CodecFloatHip &) = delete;
    CodecFloatHip &operator=(const CodecFloatHip &) = delete;

    std::optional<std::vector<uint8_t>> encode(int, int, int, const std::vector<int32_t> &) override { return std::nullopt; }
    std::optional<std::vector<uint8_t>> encodeFloats(int codecIndex, int nRows, int nCols, const std::vector<float> &values) override
    {
        std::vector<uint8_t> out(5 * values.size() + 4096);
        size_t n = 0;
        const gf_status s = gf_float_encode_f32(ctx_, codecIndex, nRows, nCols, values.data(), level_, out.data(), out.size(), &n);
        if (s < 0) throw std::runtime_error(std::string("gf_float_encode_f32: ") + gf_status_string(s));
        out.resize(n);
        return out;
    }
    bool implementsFloatingPointEncoding() const override { return true; }
    bool implementsIntegerEncoding() const override { return false; }
    std::vector<int32_t> decode(int, int, const std::vector<uint8_t> &) override { return {}; }
    std::optional<std::vector<float>> decodeFloats(int nRows, int nColumns, const std::vector<uint8_t> &packing) override
    {
        std::vector<float> out((size_t)nRows * (size_t)nColumns);
        const gf_status s = gf_float_decode_f32(ctx_, nRows, nColumns, packing.data(), packing.size(), out.data());
        if (s == GF_ERR_FORMAT || s == GF_ERR_BOUNDS) throw IOException(gf_status_string(s));
        if (s < 0) throw std::runtime_error(std::string("gf_float_decode_f32: ") + gf_status_string(s));
        return out;
    }
    void analyze(int, int, const std::vector<uint8_t> &) override {}
    void reportAnalysisData(std::FILE *ps, int) override { std::fprintf(ps, "Gridfour_Float (HIP)\n"); }
    void clearAnalysisData() override {}

private:
    gf_context *ctx_ = nullptr;
    int level_;
};

/** Drop-in for org.gridfour.compress.canonicalHuffman.CodecCanonHuffman (CodecCanonHuffman.java:70-195), the default
 *  integer codec of current Gridfour. */
class CodecCanonHuffmanHip : public ICompressionEncoder, public ICompressionDecoder {
public:
    explicit CodecCanonHuffmanHip(int device = 0)
    {
        const gf_status s = gf_context_create(device, &ctx_);
        if (s != GF_OK) throw std::runtime_error(std::string("CodecCanonHuffmanHip: ") + gf_status_string(s) + " [" + gf_last_error() + "]");
    }
    ~CodecCanonHuffmanHip() override { gf_context_destroy(ctx_); }
    CodecCanonHuffmanHip(const CodecCanonHuffmanHip &) = delete;
    CodecCanonHuffmanHip &operator=(const CodecCanonHuffmanHip &) = delete;

    std::optional<std::vector<uint8_t>> encode(int codecIndex, int nRows, int nCols, const std::vector<int32_t> &values) override
    {
        if ((size_t)nRows * (size_t)nCols != values.size()) throw std::invalid_argument("values.length != nRows*nCols");
        std::vector<uint8_t> out(gf_canon_max_packing(nRows, nCols));
        size_t n = 0;
        const gf_status s = gf_canon_encode_i32(ctx_, codecIndex, nRows, nCols, values.data(), out.data(), out.size(), &n);
        if (s == GF_DECLINED) return std::nullopt;                       // CodecCanonHuffman.java:85-87
        if (s == GF_ERR_BOUNDS) throw ArrayIndexOutOfBoundsException("PredictorModelLinear needs nCols >= 2");
        if (s == GF_ERR_ARG) throw std::invalid_argument("Empty or null data input data");   // CanonicalHuffman.java:183-185
        if (s < 0) throw std::runtime_error(std::string("gf_canon_encode_i32: ") + gf_status_string(s));
        out.resize(n);
        return out;
    }
    std::optional<std::vector<uint8_t>> encodeFloats(int, int, int, const std::vector<float> &) override { return std::nullopt; }
    bool implementsFloatingPointEncoding() const override { return false; }
    bool implementsIntegerEncoding() const override { return true; }
    std::vector<int32_t> decode(int nRows, int nColumns, const std::vector<uint8_t> &packing) override
    {
        std::vector<int32_t> out((size_t)nRows * (size_t)nColumns);
        const gf_status s = gf_canon_decode_i32(ctx_, nRows, nColumns, packing.data(), packing.size(), out.data());
        if (s == GF_ERR_FORMAT || s == GF_ERR_BOUNDS) throw IOException(gf_status_string(s));
        if (s < 0) throw std::runtime_error(std::string("gf_canon_decode_i32: ") + gf_status_string(s));
        return out;
    }
    std::optional<std::vector<float>> decodeFloats(int, int, const std::vector<uint8_t> &) override { return std::nullopt; }
    void analyze(int, int, const std::vector<uint8_t> &) override {}
    void reportAnalysisData(std::FILE *ps, int) override { std::fprintf(ps, "GVRS Canonical Huffman (HIP)\n"); }
    void clearAnalysisData() override {}

private:
    gf_context *ctx_ = nullptr;
};

/** Drop-in for org.gridfour.lsop.LsEncoder12 + LsDecoder12 (codec id "LSOP12"). */
class LsCodecHip : public ICompressionEncoder, public ICompressionDecoder {
public:
    explicit LsCodecHip(int device = 0, bool deflateEnabled = true) : deflate_(deflateEnabled)
    {
        const gf_status s = gf_context_create(device, &ctx_);
        if (s != GF_OK) throw std::runtime_error(std::string("LsCodecHip: ") + gf_status_string(s) + " [" + gf_last_error() + "]");
    }
    ~LsCodecHip() override { gf_context_destroy(ctx_); }
    LsCodecHip(const LsCodecHip &) = delete;
    LsCodecHip &operator=(const LsCodecHip &) = delete;
    void setDeflateEnabled(bool enabled) { deflate_ = enabled; }         // LsEncoder12.java:94-96
    void setValueChecksumEnabled(bool enabled) { checksum_ = enabled; }  // LsEncoder12.java:117-119

    std::optional<std::vector<uint8_t>> encode(int codecIndex, int nRows, int nCols, const std::vector<int32_t> &values) override
    {
        if ((size_t)nRows * (size_t)nCols != values.size()) throw std::invalid_argument("values.length != nRows*nCols");
        std::vector<uint8_t> out(gf_lsop12_max_packing(nRows, nCols) + 64);
        size_t n = 0;
        const gf_status s = gf_lsop12_encode_i32(ctx_, codecIndex, nRows, nCols, values.data(), (deflate_ ? GF_LSOP_DEFLATE : 0) | (checksum_ ? GF_LSOP_VALUE_CHECKSUM : 0), out.data(),
                                                 out.size(), &n);
        if (s == GF_DECLINED) return std::nullopt;                       // LsEncoder12.java:124-127
        if (s < 0) throw std::runtime_error(std::string("gf_lsop12_encode_i32: ") + gf_status_string(s));
        out.resize(n);
        return out;
    }
    std::optional<std::vector<uint8_t>> encodeFloats(int, int, int, const std::vector<float> &) override { return std::nullopt; }
    bool implementsFloatingPointEncoding() const override { return false; }
    bool implementsIntegerEncoding() const override { return true; }
    std::vector<int32_t> decode(int nRows, int nColumns, const std::vector<uint8_t> &packing) override
    {
        std::vector<int32_t> out((size_t)nRows * (size_t)nColumns);
        const gf_status s = gf_lsop12_decode_i32(ctx_, nRows, nColumns, packing.data(), packing.size(), out.data());
        if (s == GF_ERR_FORMAT || s == GF_ERR_BOUNDS) throw IOException(gf_status_string(s));
        if (s < 0) throw std::runtime_error(std::string("gf_lsop12_decode_i32: ") + gf_status_string(s));
        return out;
    }
    std::optional<std::vector<float>> decodeFloats(int, int, const std::vector<uint8_t> &) override { return std::nullopt; }
    void analyze(int, int, const std::vector<uint8_t> &) override {}
    void reportAnalysisData(std::FILE *ps, int) override { std::fprintf(ps, "LSOP12 (HIP)\n"); }
    void clearAnalysisData() override {}

private:
    gf_context *ctx_ = nullptr;
    bool deflate_;
    bool checksum_ = false;
};

/** Drop-in for org.gridfour.compress.CodecDeflate (CodecDeflate.java:108-228): predictor + CodecM32 on the GPU,
 *  Deflate (level 6) on the host's zlib. */
class CodecDeflateHip : public ICompressionEncoder, public ICompressionDecoder {
public:
    explicit CodecDeflateHip(int device = 0)
    {
        const gf_status s = gf_context_create(device, &ctx_);
        if (s != GF_OK) throw std::runtime_error(std::string("CodecDeflateHip: ") + gf_status_string(s) + " [" + gf_last_error() + "]");
    }
    ~CodecDeflateHip() override { gf_context_destroy(ctx_); }
    CodecDeflateHip(const CodecDeflateHip &) = delete;
    CodecDeflateHip &operator=(const CodecDeflateHip &) = delete;

    std::optional<std::vector<uint8_t>> encode(int codecIndex, int nRows, int nCols, const std::vector<int32_t> &values) override
    {
        if ((size_t)nRows * (size_t)nCols != values.size()) throw std::invalid_argument("values.length != nRows*nCols");
        std::vector<uint8_t> out(gf_m32_max_stream(nRows, nCols) + 256);
        size_t n = 0;
        const gf_status s = gf_deflate_encode_i32(ctx_, codecIndex, nRows, nCols, values.data(), out.data(), out.size(), &n);
        if (s == GF_DECLINED) return std::nullopt;                       // CodecDeflate.java:168-170
        if (s == GF_ERR_BOUNDS) throw ArrayIndexOutOfBoundsException("PredictorModelLinear needs nCols >= 2");
        if (s < 0) throw std::runtime_error(std::string("gf_deflate_encode_i32: ") + gf_status_string(s));
        out.resize(n);
        return out;
    }
    std::optional<std::vector<uint8_t>> encodeFloats(int, int, int, const std::vector<float> &) override { return std::nullopt; }
    bool implementsFloatingPointEncoding() const override { return false; }
    bool implementsIntegerEncoding() const override { return true; }
    std::vector<int32_t> decode(int nRows, int nColumns, const std::vector<uint8_t> &packing) override
    {
        std::vector<int32_t> out((size_t)nRows * (size_t)nColumns);
        const gf_status s = gf_deflate_decode_i32(ctx_, nRows, nColumns, packing.data(), packing.size(), out.data());
        if (s == GF_ERR_FORMAT || s == GF_ERR_BOUNDS) throw IOException(gf_status_string(s));
        if (s < 0) throw std::runtime_error(std::string("gf_deflate_decode_i32: ") + gf_status_string(s));
        return out;
    }
    std::optional<std::vector<float>> decodeFloats(int, int, const std::vector<uint8_t> &) override { return std::nullopt; }
    void analyze(int, int, const std::vector<uint8_t> &) override {}
    void reportAnalysisData(std::FILE *ps, int) override { std::fprintf(ps, "Gridfour_Deflate (HIP)\n"); }
    void clearAnalysisData() override {}

private:
    gf_context *ctx_ = nullptr;
};

}  // namespace gridfour
