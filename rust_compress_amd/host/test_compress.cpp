// test_compress.cpp -- the reference crate's own unit tests restated against compress.hpp
// (lz4.rs:647-726, flate.rs:528-582, zlib.rs:151-203, bwt/mod.rs:528-551, mtf.rs:179-197, dc.rs:259-302,
//  ari/test.rs:8-20,52-89, rle.rs:320-361).  Needs a GPU; run by tests/test_gpu_cpp_host.py.
//   g++ -std=c++17 test_compress.cpp -L../csrc -lrcx -Wl,-rpath,../csrc -o test_compress && ./test_compress <golden dir>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include "compress.hpp"

using namespace compress;
typedef std::vector<uint8_t> Bytes;
static std::string G;
static Bytes file(const std::string& name)
{
    std::ifstream f(G + "/" + name, std::ios::binary);
    if (!f) { fprintf(stderr, "missing fixture %s\n", name.c_str()); exit(2); }
    return Bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static Bytes B(const char* s) { return Bytes(s, s + strlen(s)); }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

template <class D> static Bytes one_byte_at_a_time(D& d)
{
    Bytes out; uint8_t b;
    CHECK(!d.eof());
    while (d.read(&b, 1) == 1) out.push_back(b);
    CHECK(d.eof());
    return out;
}
template <class D> static Bytes random_lengths(D& d)
{
    std::mt19937 rng(7); Bytes out; uint8_t buf[40];
    for (;;) { size_t k = d.read(buf, 1 + rng() % 40); if (!k) break; out.insert(out.end(), buf, buf + k); }
    return out;
}

int main(int argc, char** argv)
{
    G = argc > 1 ? argv[1] : "tests/golden";
    const Bytes txt = file("test.txt");
    // ---- lz4 (lz4.rs:647-726)
    for (int i = 1; i <= 9; i++) { Bytes f = file("test.lz4." + std::to_string(i)); lz4::Decoder<SliceReader> d{SliceReader(f)}; CHECK(d.read_to_end() == txt); }
    { Bytes enc, dec; lz4::encode_block(txt, enc); lz4::decode_block(enc, dec); CHECK(dec == txt); CHECK(enc.size() == 2724); }
    { Bytes f = file("test.lz4.1"); lz4::Decoder<SliceReader> d{SliceReader(f)}; CHECK(one_byte_at_a_time(d) == txt); }
    { Bytes f = file("test.lz4.1"); lz4::Decoder<SliceReader> d{SliceReader(f)}; CHECK(random_lengths(d) == txt); }
    for (const Bytes& data : {B("test"), B(""), txt}) {
        lz4::Encoder<VecWriter> e{VecWriter()};
        e.write(data.data(), data.size());
        VecWriter w = e.finish();
        lz4::Decoder<SliceReader> d{SliceReader(w.v)};
        CHECK(d.read_to_end() == data);
    }
    CHECK(!lz4::compression_bound(0x7e000001u).has_value() && *lz4::compression_bound(100) == 120);
    try { Bytes bad = B("\x01\x02\x03\x04zzzz"); lz4::Decoder<SliceReader> d{SliceReader(bad)}; d.read_to_end(); CHECK(false); }
    catch (const io_error& e) { CHECK(e.kind == ErrorKind::InvalidInput && std::string(e.what()).empty()); }
    // ---- decode_many: the frames / members of many readers through one batch call == the single decoders
    {
        std::vector<Bytes> frames; std::vector<size_t> used;
        for (int i = 1; i <= 9; i++) frames.push_back(file("test.lz4." + std::to_string(i)));
        auto outs = lz4::decode_many(frames, &used);
        CHECK(outs.size() == 9 && used.size() == 9);
        for (size_t i = 0; i < 9; i++) { lz4::Decoder<SliceReader> d{SliceReader(frames[i])}; CHECK(outs[i] == txt && d.read_to_end() == txt && d.consumed == used[i]); }
        std::vector<Bytes> zs; for (int i = 0; i < 10; i++) zs.push_back(file("test.z." + std::to_string(i)));
        auto zr = zlib::decode_many(zs);
        for (size_t i = 0; i < 10; i++) CHECK(zr.out[i] == txt && zr.consumed[i] == zs[i].size());
        Bytes bad = zs[3]; bad.back() ^= 0x55;
        int st_one = 0, st_many = 0;
        try { zlib::Decoder<SliceReader> d{SliceReader(bad)}; d.read_to_end(); } catch (const io_error& e) { st_one = e.status; }
        try { zlib::decode_many({zs[0], bad, zs[1]}); } catch (const io_error& e) { st_many = e.status; }
        CHECK(st_one != 0 && st_one == st_many);
        CHECK(flate::decode_many({}).out.empty());
    }
    // ---- a batch sharded over several contexts (here: the one GPU, named three times): block for block what one context returns
    {
        std::vector<Bytes> raws, encs; std::vector<uint64_t> caps;
        for (int i = 0; i < 23; i++) { Bytes r(txt.begin(), txt.begin() + (txt.size() * (i + 1)) / 23); raws.push_back(r); caps.push_back(*lz4::compression_bound((uint32_t)r.size())); }
        auto call_e = [](rcx_ctx* c, rcx_batch* b, uint32_t*) { return rcx_lz4_encode_batch(c, b); };
        BatchResult one = run_batch(raws, caps, call_e), many = run_batch_devices({0, 0, 0}, raws, caps, call_e);
        CHECK(one.out == many.out && one.status == many.status && one.in_used == many.in_used);
        std::vector<uint64_t> dcaps; for (auto& r : raws) dcaps.push_back(r.size());
        BatchResult dec = run_batch_devices({0, 0}, many.out, dcaps, [](rcx_ctx* c, rcx_batch* b, uint32_t*) { return rcx_lz4_decode_batch(c, b); });
        CHECK(dec.out == raws);
        std::vector<size_t> bd = partition(dcaps, 3);
        CHECK(bd.size() == 4 && bd[0] == 0 && bd[3] == raws.size() && bd[1] > 0 && bd[2] > bd[1] && bd[2] < raws.size());
        // dc::Context from both sides (dc.rs:268-289: the decoder is handed the contexts the encoder yields)
        bwt::dc::Encoded e = bwt::dc::encode(txt);
        std::vector<uint32_t> ds; for (auto& p : e.pairs) ds.push_back(p.first);
        auto back = bwt::dc::decode(e.init, ds, txt.size());
        CHECK(back.first == txt && back.second.size() == e.pairs.size());
        for (size_t j = 0; j < e.pairs.size(); j++) CHECK(back.second[j] == e.pairs[j].second);
    }
    // ---- flate / zlib (flate.rs:528-582, zlib.rs:151-203)
    for (int i = 0; i <= 9; i++) {
        Bytes z = file("test.z." + std::to_string(i));
        { zlib::Decoder<SliceReader> d{SliceReader(z)}; CHECK(d.read_to_end() == txt); }
        Bytes raw(z.begin() + 2, z.end() - 4);                                       // fixup, flate.rs:504-506
        { flate::Decoder<SliceReader> d{SliceReader(raw)}; CHECK(d.read_to_end() == txt); }
    }
    { Bytes z = file("test.z.go"); flate::Decoder<SliceReader> d{SliceReader(z)}; CHECK(d.read_to_end() == txt); CHECK(d.flags == RCX_W_EMPTY_BLOCK_MIDSTREAM); }
    { Bytes z = file("test.z.1"); zlib::Decoder<SliceReader> d{SliceReader(z)}; CHECK(one_byte_at_a_time(d) == txt); }
    { Bytes z = file("test.z.1"); Bytes raw(z.begin() + 2, z.end() - 4); flate::Decoder<SliceReader> d{SliceReader(raw)}; CHECK(random_lengths(d) == txt); }
    try { Bytes z = file("test.z.1"); z.back() ^= 1; zlib::Decoder<SliceReader> d{SliceReader(z)}; d.read_to_end(); CHECK(false); }
    catch (const io_error& e) { CHECK(e.kind == ErrorKind::InvalidInput && std::string(e.what()) == "invalid checksum on zlib stream"); }
    { Adler32 a; a.feed((const uint8_t*)"abra", 4); a.feed((const uint8_t*)"cadabra", 7); CHECK(a.result() == 0x19f20455u); }
    // ---- bwt / mtf / dc
    for (const Bytes& data : {B("abracadabra"), txt}) {
        bwt::Encoder<VecWriter> e(VecWriter(), 1024);
        e.write(data.data(), data.size());
        VecWriter w = e.finish();
        bwt::Decoder<SliceReader> d(SliceReader(w.v), true);
        CHECK(d.read_to_end() == data);
        auto enc = bwt::encode_simple(data);
        CHECK(bwt::decode_simple(enc.first, enc.second) == data);
        CHECK(bwt::mtf::decode(bwt::mtf::encode(data)) == data);
    }
    { auto e = bwt::encode_simple(B("abracadabra")); CHECK(e.first == B("rdarcaaaabb") && e.second == 2); }
    {   // compute_suffixes / compute_inversion_table, bwt/mod.rs:136-166, 223-239: "abracadabra" by hand
        std::vector<uint32_t> sa(11), tab(11);
        bwt::compute_suffixes(B("abracadabra"), sa);
        CHECK((sa == std::vector<uint32_t>{10, 7, 0, 3, 5, 8, 1, 4, 6, 9, 2}));
        bwt::compute_inversion_table(B("rdarcaaaabb"), 2, tab);        // place(): a -> 0.., b -> 5.., c -> 7, d -> 8, r -> 9..; origin (an 'a') first
        CHECK((tab == std::vector<uint32_t>{0, 6, 7, 8, 9, 10, 11, 5, 2, 1, 4}));
        // the table drives InverseIterator::next (:266-281)
        std::vector<uint8_t> back; size_t cur = 2; const auto L = B("rdarcaaaabb");
        for (int k = 0; k < 11; k++) { cur = (size_t)tab[cur] - 1; const size_t p = cur != (size_t)-1 ? cur : 2; back.push_back(L[p]); if (cur == (size_t)-1) break; }
        CHECK(back == B("abracadabra"));
        bool threw = false;
        try { std::vector<uint32_t> shortt(10); bwt::compute_inversion_table(B("rdarcaaaabb"), 2, shortt); } catch (const io_error&) { threw = true; }
        CHECK(threw);
    }
    for (const char* s : {"abracadabra", "banana", "test"}) {                         // extra_mem = false: decode_minimal, bwt/mod.rs:298-315, 549-551
        bwt::Encoder<VecWriter> e(VecWriter(), 64);
        e.write((const uint8_t*)s, strlen(s));
        VecWriter w = e.finish();
        bwt::Decoder<SliceReader> d(SliceReader(w.v), false);
        const bool same = d.read_to_end() == B(s);
        CHECK(same == (std::string(s) != "test"));                                    // the reference's function is not an inverse on "test" 
    }
    for (const Bytes& data : {B("teeesst_dc"), B(""), txt}) {
        auto d = bwt::dc::encode_simple(data);
        CHECK(bwt::dc::decode_simple(data.size(), d) == data);
    }
    { auto d = bwt::dc::encode_simple(B("teeesst_dc")); CHECK(d.size() == 263 && d[256] == 3 && d[257] == 1 && d['t'] == 0 && d['e'] == 1); }
    // ---- ari (test.rs:8-20, 52-89)
    for (const Bytes& data : {B("abracadabra"), B(""), txt}) {
        entropy::ari::ByteEncoder<VecWriter> e{VecWriter()};
        e.write(data.data(), data.size());
        VecWriter w = e.finish();
        entropy::ari::ByteDecoder<SliceReader> d{SliceReader(w.v)};
        CHECK(d.read_to_end() == data);
    }
    {   // roundtrip_term: two terminated streams back to back
        VecWriter w;
        for (const Bytes& part : {B("abra"), B("cadabra")}) { entropy::ari::ByteEncoder<VecWriter> e{std::move(w)}; e.write(part.data(), part.size()); w = e.finish(); }
        entropy::ari::ByteDecoder<SliceReader> d1{SliceReader(w.v)};
        CHECK(d1.read_to_end() == B("abra"));
        // the second decoder continues on the SAME reader, left exactly after the first stream (ari/test.rs:52-89)
        entropy::ari::ByteDecoder<TailReader<SliceReader>> d2{std::move(d1.finish())};
        CHECK(d2.read_to_end() == B("cadabra"));
        uint8_t dummy;
        CHECK(d2.finish().read(&dummy, 1) == 0);
    }
    // ---- rle (rle.rs:320-361)
    auto renc = [](const Bytes& b) { rle::Encoder<VecWriter> e{VecWriter()}; e.write_all(b.data(), b.size()); return e.finish().v; };
    auto rdec = [](const Bytes& b) { rle::Decoder<SliceReader> d{SliceReader(b)}; return d.read_to_end(); };
    CHECK(renc(B("")) == B("") && renc(B("a")) == B("a") && renc(B("abca123")) == B("abca123"));
    CHECK(renc(Bytes{20, 20, 20, 20, 20, 15}) == (Bytes{20, 20, 131, 15}));
    CHECK(renc(Bytes(129, 5)) == (Bytes{5, 5, 255}));
    CHECK(rdec(Bytes{20, 20, 131, 15}) == (Bytes{20, 20, 20, 20, 20, 15}));
    { std::mt19937 rng(1); for (int it = 0; it < 10; it++) { Bytes b(13579); for (auto& x : b) x = (uint8_t)rng(); CHECK(rdec(renc(b)) == b); } }
    try { Bytes bad = {7, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; rdec(bad); CHECK(false); }
    catch (const io_error& e) { CHECK(std::string(e.what()) == "Overly long run"); }
    printf("CPP_HOST_OK\n");
    return 0;
}
