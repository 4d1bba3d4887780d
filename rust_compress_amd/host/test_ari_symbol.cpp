// test_ari_symbol.cpp -- the per-symbol range-coder surface (ari_symbol.hpp) driven exactly as the reference's tests drive
// theirs (src/entropy/ari/test.rs:22-50 encode_binary / roundtrip_binary, :91-148 roundtrip_proxy, :150-182 roundtrip_apm,
// and the byte model of table.rs:185-273 through the generic Encoder / Decoder).  Pure host code: runs without a GPU.
//   usage: test_ari_symbol <input file> <rate>   ->   four lines "name hex", after every stream has round-tripped
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include "ari_symbol.hpp"
using namespace compress::entropy;
typedef std::vector<uint8_t> Bytes;
struct VecWriter { Bytes v; void write(const uint8_t* p, size_t n) { v.insert(v.end(), p, p + n); } };
struct SliceReader { const Bytes* d; size_t pos = 0; explicit SliceReader(const Bytes& b) : d(&b) {} size_t read(uint8_t* dst, size_t n) { size_t k = 0; while (k < n && pos < d->size()) dst[k++] = (*d)[pos++]; return k; } };
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); exit(1); } } while (0)
static void hex(const char* name, const Bytes& b) { printf("%s ", name); for (uint8_t x : b) printf("%02x", x); printf("\n"); }

static Bytes encode_binary(const Bytes& in, ari::bin::Model& model)                  // test.rs:22-38
{
    ari::Encoder<VecWriter> e{VecWriter()};
    for (uint8_t byte : in) for (int i = 0; i < 8; i++) { const bool bit = (byte >> i) & 1; e.encode(bit, model); model.update(bit); }
    return e.finish().v;
}
int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    const Bytes in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const uint32_t rate = (uint32_t)atoi(argv[2]);
    {   // roundtrip_binary, test.rs:40-50
        ari::bin::Model bm = ari::bin::Model::new_flat(ari::RANGE_DEFAULT_THRESHOLD >> 3, rate);
        const Bytes out = encode_binary(in, bm);
        bm.reset_flat();
        ari::Decoder<SliceReader> d{SliceReader(out)};
        for (uint8_t byte : in) { uint8_t v = 0; for (int i = 0; i < 8; i++) { const bool bit = d.decode(bm); bm.update(bit); v = (uint8_t)(v + ((uint8_t)bit << i)); } CHECK(v == byte); }
        hex("binary", out);
    }
    {   // roundtrip_proxy, test.rs:91-148: high nibble through two summed tables, low nibble bit by bit through two summed binary models
        const ari::Border th = ari::RANGE_DEFAULT_THRESHOLD >> 3;
        auto t0 = ari::table::Model::new_flat(16, th), t1 = ari::table::Model::new_flat(16, th);
        auto b0 = ari::bin::Model::new_flat(th, 3), b1 = ari::bin::Model::new_flat(th, 5);
        ari::Encoder<VecWriter> e{VecWriter()};
        for (uint8_t byte : in) {
            const size_t high = byte >> 4;
            e.encode(high, ari::table::SumProxy(2, t0, 1, t1, 0));
            t0.update(high, 10, 1); t1.update(high, 5, 1);
            for (int i = 0; i < 4; i++) { const bool bit = (byte >> i) & 1; e.encode(bit, ari::bin::SumProxy(1, b0, 1, b1, 1)); b0.update(bit); b1.update(bit); }
        }
        const Bytes out = e.finish().v;
        t0.reset_flat(); t1.reset_flat(); b0.reset_flat(); b1.reset_flat();
        ari::Decoder<SliceReader> d{SliceReader(out)};
        for (uint8_t byte : in) {
            const size_t high = d.decode(ari::table::SumProxy(2, t0, 1, t1, 0));
            t0.update(high, 10, 1); t1.update(high, 5, 1);
            uint8_t v = (uint8_t)(high << 4);
            for (int i = 0; i < 4; i++) { const bool bit = d.decode(ari::bin::SumProxy(1, b0, 1, b1, 1)); v = (uint8_t)(v + ((uint8_t)bit << i)); b0.update(bit); b1.update(bit); }
            CHECK(v == byte);
        }
        hex("proxy", out);
    }
    {   // roundtrip_apm, test.rs:150-182 (a bit history that leaves the gate's bins panics in the reference: reported, not a failure)
        try {
            auto bit = ari::apm::Bit::new_equal(); ari::apm::Gate gate;
            ari::Encoder<VecWriter> e{VecWriter()};
            for (uint8_t b8 : in) for (int i = 0; i < 8; i++) {
                const bool b1 = (b8 >> i) & 1;
                const auto p = gate.pass(bit);
                e.encode(b1, p.first);
                bit.update(b1, 10, 0); gate.update(b1, p.second, 10, 0);
            }
            const Bytes out = e.finish().v;
            bit = ari::apm::Bit::new_equal(); gate = ari::apm::Gate();
            ari::Decoder<SliceReader> d{SliceReader(out)};
            for (uint8_t b8 : in) {
                uint8_t v = 0;
                for (int i = 0; i < 8; i++) { const auto p = gate.pass(bit); const bool b1 = d.decode(p.first); if (b1) v = (uint8_t)(v + (1u << i)); bit.update(b1, 10, 0); gate.update(b1, p.second, 10, 0); }
                CHECK(v == b8);
            }
            hex("apm", out);
        } catch (const ari::panic_error&) { printf("apm panic\n"); }
    }
    {   // the byte model of ByteEncoder / ByteDecoder (table.rs:185-273) through the generic coder: 257 values, the last one ends the stream
        const ari::Border fmax = ari::RANGE_DEFAULT_THRESHOLD >> 2;
        auto freq = ari::table::Model::new_flat(257, fmax);
        ari::Encoder<VecWriter> e{VecWriter()};
        for (uint8_t b : in) { e.encode((size_t)b, freq); freq.update(b, 10, 1); }
        e.encode((size_t)256, freq);
        Bytes out = e.finish().v;
        out.push_back('x'); out.push_back('y');                                       // the reader must be left right behind the stream
        freq = ari::table::Model::new_flat(257, fmax);
        ari::Decoder<SliceReader> d{SliceReader(out)};
        Bytes back;
        for (;;) { const size_t v = d.decode(freq); if (v == 256) break; freq.update(v, 10, 1); back.push_back((uint8_t)v); }
        CHECK(back == in);
        SliceReader rest = d.finish();
        CHECK(rest.pos == out.size() - 2);
        out.resize(out.size() - 2);
        hex("byte", out);
    }
    {   // RangeEncoder on its own, mod.rs:117-169
        ari::RangeEncoder re(ari::RANGE_DEFAULT_THRESHOLD);
        uint8_t o[8];
        CHECK(re.process(4, 1, 2, o) == 0 && re.low() == 0x3fffffffu && re.hai() == 0x7ffffffeu);
        CHECK(re.query(4, 0x50000000u) == 1);
        re.reset();
        CHECK(re.low() == 0 && re.hai() == 0xffffffffu);
        CHECK(re.process(256, 65, 66, o) == 0 && re.low() == 0x40ffffbfu && re.hai() == 0x41ffffbeu);      // a 1/256 share: the top bytes still differ
        CHECK(re.process(0xffff, 0, 1, o) == 3 && o[0] == 0x41 && o[1] == 0 && o[2] == 0);                  // 256 wide, across a byte boundary: cut at 0x41000000 (the larger side), three bytes leave
        CHECK(re.low() == 0 && re.hai() == 0xbf000000u);
        const ari::Border lo = re.low();
        CHECK(re.get_code_tail() == lo && re.low() == 0 && re.hai() == 0);
    }
    {   // a zero-width interval panics (the reference: assert / output[4] out of bounds) instead of shipping bytes for ever
        auto panics = [](auto&& f) { try { f(); } catch (const ari::panic_error&) { return true; } return false; };
        CHECK(panics([] { ari::Encoder<VecWriter> e{VecWriter()}; e.encode(false, ari::bin::Model::new_custom(0, 1u << 11, 5)); }));
        CHECK(panics([] { ari::RangeEncoder re(ari::RANGE_DEFAULT_THRESHOLD); uint8_t o[8]; re.process(10, 3, 3, o); }));
        CHECK(panics([] { ari::RangeEncoder re(ari::RANGE_DEFAULT_THRESHOLD); uint8_t o[8]; re.process(10, 3, 11, o); }));
        CHECK(panics([] { ari::RangeEncoder re(ari::RANGE_DEFAULT_THRESHOLD); uint8_t o[8]; re.process(0, 0, 0, o); }));
    }
    printf("ARI_SYMBOL_OK\n");
    return 0;
}
