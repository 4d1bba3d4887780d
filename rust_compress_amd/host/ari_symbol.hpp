// ari_symbol.hpp -- the reference's PER-SYMBOL range-coder surface, as host code (C++17, no device, no librcx):
//
//   compress::entropy::ari::RangeEncoder                      src/entropy/ari/mod.rs:67-169
//   compress::entropy::ari::Model<Derived, V>  (the trait)    src/entropy/ari/mod.rs:174-204
//   compress::entropy::ari::Encoder<W> / Decoder<R>           src/entropy/ari/mod.rs:208-293
//   compress::entropy::ari::table::{Model, SumProxy}          src/entropy/ari/table.rs:20-180
//   compress::entropy::ari::bin::{Model, SumProxy}            src/entropy/ari/bin.rs:17-167
//   compress::entropy::ari::apm::{Bit, Gate}                  src/entropy/ari/apm.rs:36-198
//
// Why host code: this surface codes ONE decision per call against a model the caller owns and mutates between calls
// (test.rs:22-50, 91-148) -- there is no batch to hand a device.  The whole-stream codecs over the same arithmetic
// (ByteEncoder / ByteDecoder and the encode_bytes forms) are the device's, in compress.hpp; the bytes agree, which
// tests/test_ari_symbol_host.py checks symbol by symbol against the oracle's streams.
//
// A writer W has `void write(const uint8_t*, size_t)`; a reader R has `size_t read(uint8_t*, size_t)` (0 at the end).
// A Rust panic (assert!, index out of bounds, unwrap on None / Err) is a `panic_error`.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace compress { namespace entropy { namespace ari {

struct panic_error : std::logic_error { using std::logic_error::logic_error; };

typedef uint8_t Symbol;
typedef uint32_t Border;
constexpr Border RANGE_DEFAULT_THRESHOLD = 1u << 14;            // mod.rs:61
constexpr size_t BORDER_BYTES = 4;

// ---- mod.rs:67-169 ------------------------------------------------------------------------------------------------
// The interval [low, hai) lives in 32 bits; a symbol narrows it to its share and every leading byte both ends agree on is
// shipped.  When the ends straddle a byte boundary but are closer than `threshold`, the interval is cut at the boundary
// (the larger side survives) so that a byte can leave: precision is never allowed to fall below `threshold`.
class RangeEncoder {
public:
    Border threshold;
    explicit RangeEncoder(Border max_range) : threshold(max_range) {}
    void reset() { low_ = 0; hai_ = ~0u; }
    // [from/total, to/total) of the current interval; the bytes that leave go to output[0..], their number is returned
    size_t process(Border total, Border from, Border to, Symbol* output)
    {
        // mod.rs:118-122: `from < to && to <= total` and `range > 0` are the reference's asserts; a zero-width interval would
        // otherwise ship bytes for ever (the reference's release build dies on output[4], the slice has BORDER_BYTES entries)
        if (total == 0) throw panic_error("attempt to divide by zero (RangeEncoder::process: total == 0)");
        if (!(from < to && to <= total)) throw panic_error("assertion failed: from<to && to<=total");
        const Border width = (hai_ - low_) / total;
        if (width == 0) throw panic_error("RangeCoder range is too narrow for the total");
        Border a = low_ + width * from, b = low_ + width * to;
        size_t shipped = 0;
        for (;;) {
            if (((a ^ b) & TOP_BYTE) != 0) {                     // the top bytes differ: nothing leaves unless the interval is too narrow
                if (b - a > threshold) break;
                const Border edge = b & TOP_BYTE;
                if (b - edge >= edge - a) a = edge; else b = edge - 1;
            }
            if (shipped == BORDER_BYTES) throw panic_error("index out of bounds: the len is 4 but the index is 4 (RangeEncoder::process)");
            output[shipped++] = (Symbol)(a >> 24);
            a <<= 8; b <<= 8;
        }
        low_ = a; hai_ = b;
        return shipped;
    }
    // which offset in [0, total) `code` stands for
    Border query(Border total, Border code) const { return (code - low_) / ((hai_ - low_) / total); }
    Border get_code_tail() { const Border t = low_; low_ = 0; hai_ = 0; return t; }
    Border low() const { return low_; }
    Border hai() const { return hai_; }
private:
    static constexpr Border TOP_BYTE = 0xff000000u;
    Border low_ = 0, hai_ = ~0u;
};

// ---- the Model trait, mod.rs:174-204 --------------------------------------------------------------------------------
// A model supplies get_range(value) -> (lo, hi), find_value(offset) -> (value, lo, hi) and get_denominator(); the two
// provided methods are the trait's defaults.  (CRTP: `struct M : Model<M, V>`.)
template <class Derived, class V>
struct Model {
    typedef V value_type;
    size_t encode(V value, RangeEncoder& re, Symbol* out) const
    {
        const Derived& m = static_cast<const Derived&>(*this);
        const std::pair<Border, Border> r = m.get_range(value);
        return re.process(m.get_denominator(), r.first, r.second, out);
    }
    std::pair<V, size_t> decode(Border code, RangeEncoder& re) const
    {
        const Derived& m = static_cast<const Derived&>(*this);
        const Border total = m.get_denominator();
        const std::tuple<V, Border, Border> f = m.find_value(re.query(total, code));
        Symbol scratch[BORDER_BYTES + 4];
        const size_t shift = re.process(total, std::get<1>(f), std::get<2>(f), scratch);
        return {std::get<0>(f), shift};
    }
};

// ---- mod.rs:208-251 -------------------------------------------------------------------------------------------------
template <class W>
class Encoder {
public:
    explicit Encoder(W w) : stream_(std::move(w)), range_(RANGE_DEFAULT_THRESHOLD) {}
    template <class M>
    void encode(typename M::value_type value, const M& model)
    {
        Symbol buf[BORDER_BYTES + 4];
        const size_t n = model.encode(value, range_, buf);
        if (n) stream_.write(buf, n);
    }
    // the four bytes of the code tail, most significant first; the writer comes back
    W finish()
    {
        const Border t = range_.get_code_tail();
        const Symbol tail[4] = {(Symbol)(t >> 24), (Symbol)(t >> 16), (Symbol)(t >> 8), (Symbol)t};
        stream_.write(tail, 4);
        return std::move(stream_);
    }
    void flush() {}
    W& get_ref() { return stream_; }
private:
    W stream_; RangeEncoder range_;
};

// ---- mod.rs:254-293 -------------------------------------------------------------------------------------------------
template <class R>
class Decoder {
public:
    explicit Decoder(R r) : stream_(std::move(r)), range_(RANGE_DEFAULT_THRESHOLD) {}
    template <class M>
    typename M::value_type decode(const M& model)
    {
        if (!feed()) throw panic_error("ari::Decoder::decode: the stream ended inside the code (feed().unwrap())");
        const auto r = model.decode(code_, range_);
        pending_ = r.second;
        return r.first;
    }
    // reads what the last symbol left pending, so that the reader stands right behind the stream; false: it ended early
    bool finish_feed() { return feed(); }
    R finish() { (void)feed(); return std::move(stream_); }
    R& get_ref() { return stream_; }
private:
    bool feed()
    {
        while (pending_) {
            uint8_t b;
            if (stream_.read(&b, 1) != 1) return false;
            code_ = (code_ << 8) + b;
            pending_--;
        }
        return true;
    }
    R stream_; RangeEncoder range_; Border code_ = 0; size_t pending_ = BORDER_BYTES;
};

// ---- table.rs:20-180 ------------------------------------------------------------------------------------------------
namespace table {
typedef uint16_t Frequency;

class Model : public ari::Model<Model, size_t> {
public:
    template <class F>
    static Model new_custom(size_t num_values, Border threshold, F init)
    {
        Model m;
        m.cut_threshold_ = threshold;
        m.table_.resize(num_values);
        for (size_t i = 0; i < num_values; i++) { m.table_[i] = (Frequency)init(i); m.total_ += m.table_[i]; }
        while (m.total_ >= threshold) m.downscale();
        return m;
    }
    static Model new_flat(size_t num_values, Border threshold) { return new_custom(num_values, threshold, [](size_t) { return 1; }); }
    void reset_flat() { for (Frequency& f : table_) f = 1; total_ = (Border)table_.size(); }
    // value's frequency grows by total >> add_log plus add_const; the table is halved whenever the sum reaches the threshold
    void update(size_t value, size_t add_log, Border add_const)
    {
        const Border add = (total_ >> add_log) + add_const;
        if (!(add < 2 * cut_threshold_)) throw panic_error("table::Model::update: add >= 2 * cut_threshold");
        table_.at(value) = (Frequency)(table_.at(value) + (Frequency)add);
        total_ += add;
        if (total_ >= cut_threshold_) {
            downscale();
            if (!(total_ < cut_threshold_)) throw panic_error("table::Model::update: total >= cut_threshold after the downscale");
        }
    }
    void downscale()
    {
        const Frequency roundup = (Frequency)((1u << cut_shift_) - 1u);      // non-zero frequencies stay positive
        total_ = 0;
        for (Frequency& f : table_) { f = (Frequency)((Frequency)(f + roundup) >> cut_shift_); total_ += f; }
    }
    const std::vector<Frequency>& get_frequencies() const { return table_; }
    // the trait
    std::pair<Border, Border> get_range(size_t value) const
    {
        if (value >= table_.size()) throw panic_error("table::Model::get_range: value out of range");
        Border lo = 0;
        for (size_t i = 0; i < value; i++) lo += table_[i];
        return {lo, lo + table_[value]};
    }
    std::tuple<size_t, Border, Border> find_value(Border offset) const
    {
        if (!(offset < total_)) throw panic_error("Invalid frequency offset " + std::to_string(offset) + " requested under total " + std::to_string(total_));
        size_t v = 0; Border lo = 0, hi = table_[0];
        while (hi <= offset) { lo = hi; hi += table_.at(++v); }
        return {v, lo, hi};
    }
    Border get_denominator() const { return total_; }
private:
    Border total_ = 0; std::vector<Frequency> table_; Border cut_threshold_ = 0; size_t cut_shift_ = 1;
};

// (wa * A + wb * B) >> ws over two tables of the same size
class SumProxy : public ari::Model<SumProxy, size_t> {
public:
    SumProxy(Border wa, const table::Model& fa, Border wb, const table::Model& fb, Border shift) : a_(fa), b_(fb), wa_(wa), wb_(wb), ws_(shift)
    { if (fa.get_frequencies().size() != fb.get_frequencies().size()) throw panic_error("table::SumProxy::new: the tables differ in size"); }
    std::pair<Border, Border> get_range(size_t value) const
    {
        const auto ra = a_.get_range(value), rb = b_.get_range(value);
        return {(wa_ * ra.first + wb_ * rb.first) >> ws_, (wa_ * ra.second + wb_ * rb.second) >> ws_};
    }
    std::tuple<size_t, Border, Border> find_value(Border offset) const
    {
        const Border total = get_denominator();
        if (!(offset < total)) throw panic_error("Invalid frequency offset " + std::to_string(offset) + " requested under total " + std::to_string(total));
        const auto& fa = a_.get_frequencies(); const auto& fb = b_.get_frequencies();
        size_t v = 0; Border lo = 0, hi = 0;
        for (;;) {
            hi = lo + ((wa_ * fa.at(v) + wb_ * fb.at(v)) >> ws_);
            if (hi > offset) break;
            lo = hi; v++;
        }
        return {v, lo, hi};
    }
    Border get_denominator() const { return (wa_ * a_.get_denominator() + wb_ * b_.get_denominator()) >> ws_; }
private:
    const table::Model& a_; const table::Model& b_; Border wa_, wb_, ws_;
};
}  // namespace table

// ---- bin.rs:17-167 --------------------------------------------------------------------------------------------------
namespace bin {
class Model : public ari::Model<Model, bool> {
public:
    Border rate;
    static Model new_flat(Border threshold, Border rate) { return Model(threshold >> 1, threshold, rate); }
    static Model new_custom(uint8_t zero_percent, Border threshold, Border rate)
    {
        if (threshold < 100) throw panic_error("bin::Model::new_custom: threshold < 100");
        return Model((Border)zero_percent * threshold / 100, threshold, rate);
    }
    void reset_flat() { zero_ = total_ >> 1; }
    Border get_probability_zero() const { return zero_; }
    Border get_probability_one() const { return total_ - zero_; }
    void update_zero() { zero_ += (total_ - zero_) >> rate; }
    void update_one() { zero_ -= zero_ >> rate; }
    void update(bool value) { if (value) update_one(); else update_zero(); }
    std::pair<Border, Border> get_range(bool value) const { return value ? std::make_pair(zero_, total_) : std::make_pair((Border)0, zero_); }
    std::tuple<bool, Border, Border> find_value(Border offset) const
    {
        if (!(offset < total_)) throw panic_error("Invalid frequency offset " + std::to_string(offset) + " requested under total " + std::to_string(total_));
        if (offset < zero_) return {false, 0, zero_};
        return {true, zero_, total_};
    }
    Border get_denominator() const { return total_; }
private:
    Model(Border z, Border t, Border r) : rate(r), zero_(z), total_(t) {}
    Border zero_, total_;
};

class SumProxy : public ari::Model<SumProxy, bool> {
public:
    SumProxy(Border wa, const bin::Model& first, Border wb, const bin::Model& second, Border shift) : a_(first), b_(second), wa_(wa), wb_(wb), ws_(shift) {}
    std::pair<Border, Border> get_range(bool value) const
    {
        const Border z = zero();
        return value ? std::make_pair(z, get_denominator()) : std::make_pair((Border)0, z);
    }
    std::tuple<bool, Border, Border> find_value(Border offset) const
    {
        const Border z = zero(), total = get_denominator();
        if (!(offset < total)) throw panic_error("Invalid frequency offset " + std::to_string(offset) + " requested under total " + std::to_string(total));
        if (offset < z) return {false, 0, z};
        return {true, z, total};
    }
    Border get_denominator() const { return (wa_ * a_.get_denominator() + wb_ * b_.get_denominator()) >> ws_; }
private:
    Border zero() const { return (wa_ * a_.get_probability_zero() + wb_ * b_.get_probability_zero()) >> ws_; }
    const bin::Model& a_; const bin::Model& b_; Border wa_, wb_, ws_;
};
}  // namespace bin

// ---- apm.rs:36-198 --------------------------------------------------------------------------------------------------
// 12-bit "flat" probabilities and their stretched ("wide", ln(p / (1 - p)) scaled by 2048) form.  The two f32 functions are
// libm's logf / expf, which is what Rust's f32::ln / f32::exp call.
namespace apm {
typedef uint16_t FlatProbability;
typedef int16_t WideProbability;
constexpr int FLAT_TOTAL = 1 << 12;
constexpr int WIDE_OFFSET = 1 << 11;
constexpr size_t PORTAL_OFFSET = 8, PORTAL_BINS = 17;

class Bit : public ari::Model<Bit, bool> {
public:
    static Bit new_equal() { return Bit((FlatProbability)(FLAT_TOTAL >> 1)); }
    static Bit from_flat(FlatProbability fp) { return Bit(fp); }
    static Bit from_wide(WideProbability wp)
    {
        const float d = (float)wp / (float)WIDE_OFFSET;
        const float p = 1.0f / (1.0f + std::exp(-d));
        return Bit(to_u16(p * (float)FLAT_TOTAL));
    }
    FlatProbability to_flat() const { return fp_; }
    WideProbability to_wide() const
    {
        const float p = (float)fp_ / (float)FLAT_TOTAL;
        const float d = std::log(p / (1.0f - p));
        return to_i16(d * (float)WIDE_OFFSET);
    }
    void update_zero(long rate, long bias) { fp_ = (FlatProbability)(fp_ + (FlatProbability)(((long)FLAT_TOTAL - bias - (long)fp_) >> rate)); }
    void update_one(long rate, long bias) { fp_ = (FlatProbability)(fp_ - (FlatProbability)(((long)fp_ - bias) >> rate)); }
    void update(bool value, long rate, long bias) { if (!value) update_zero(rate, bias); else update_one(rate, bias); }
    std::pair<Border, Border> get_range(bool value) const { return value ? std::make_pair((Border)fp_, (Border)FLAT_TOTAL) : std::make_pair((Border)0, (Border)fp_); }
    std::tuple<bool, Border, Border> find_value(Border offset) const
    {
        if (!(offset < (Border)FLAT_TOTAL)) throw panic_error("Invalid bit offset " + std::to_string(offset) + " requested");
        if (offset < fp_) return {false, 0, fp_};
        return {true, fp_, (Border)FLAT_TOTAL};
    }
    Border get_denominator() const { return (Border)FLAT_TOTAL; }
private:
    explicit Bit(FlatProbability fp) : fp_(fp) {}
    // num's ToPrimitive for f32 -> integer: None (and the unwrap panics) unless the value lies strictly inside the target's
    // range extended by one on either side, then truncation towards zero (the reading oracle/o_ari.c pins)
    static WideProbability to_i16(float x)
    {
        if (!(x > -32769.0f && x < 32768.0f)) throw panic_error("to_i16().unwrap() on a value out of range");
        return (WideProbability)x;
    }
    static FlatProbability to_u16(float x)
    {
        if (!(x > -1.0f && x < 65536.0f)) throw panic_error("to_u16().unwrap() on a value out of range");
        return (FlatProbability)x;
    }
    FlatProbability fp_;
};

typedef std::pair<size_t, size_t> BinCoords;                     // (index, weight)

class Gate {
public:
    Gate()
    {
        for (size_t i = 0; i < PORTAL_BINS; i++) {
            const float rp = (float)i / (float)PORTAL_OFFSET - 1.0f;
            map_.push_back(Bit::from_wide((WideProbability)(rp * (float)WIDE_OFFSET)));
        }
    }
    std::pair<Bit, BinCoords> pass(const Bit& bit) const
    {
        const auto r = pass_wide(bit.to_wide());
        return {Bit::from_flat(r.first), r.second};
    }
    std::pair<FlatProbability, BinCoords> pass_wide(WideProbability wp) const
    {
        const long idx = ((long)wp + WIDE_OFFSET) >> 8;
        if (idx < 0 || (size_t)idx + 1 >= PORTAL_BINS) throw panic_error("apm::Gate::pass_wide: bin index out of bounds");
        const size_t weight = (size_t)(uint16_t)wp & 255u;
        const size_t sum = (size_t)map_[idx].to_flat() * (256 - weight) + (size_t)map_[idx + 1].to_flat() * weight;
        return {(FlatProbability)(sum >> 8), {(size_t)idx, weight}};
    }
    void update_zero(BinCoords bc, long rate, long bias) { map_.at(bc.first).update_zero(rate, bias); map_.at(bc.first + 1).update_zero(rate, bias); }
    void update_one(BinCoords bc, long rate, long bias) { map_.at(bc.first).update_one(rate, bias); map_.at(bc.first + 1).update_one(rate, bias); }
    void update(bool value, BinCoords bc, long rate, long bias) { if (!value) update_zero(bc, rate, bias); else update_one(bc, rate, bias); }
private:
    std::vector<Bit> map_;
};
}  // namespace apm

}}}  // namespace compress::entropy::ari
