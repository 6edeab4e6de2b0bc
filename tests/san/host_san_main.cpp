// host_san_main.cpp -- AddressSanitizer / UndefinedBehaviorSanitizer target for the host-only C++ of libsdmi (SURVEY.md
// section 5 promised a sanitizer test build; the reference has none).  Built by tests/test_sanitizers_cpu.py with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// from csrc/tokenizer.cpp, csrc/png_writer.cpp, csrc/mpk_reader.cpp (no HIP anywhere in these).
//
//   host_san <merges file> <record.mpk> <scratch dir>
//
// * tokenizer: encode / decode of awkward strings (invalid UTF-8, long runs, every byte value), round trips;
// * PNG writer: a few image sizes incl. 1x1 and a stored-deflate block boundary;
// * .mpk reader: the good record, then EVERY truncation of it and 3000 single-byte corruptions -- each must either parse
//   or throw sdmi::Error; any out-of-bounds read, overflow or leak aborts the process (non-zero exit).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../stable_diffusion_burn_amd/csrc/error.hpp"
#include "../../stable_diffusion_burn_amd/csrc/mpk_reader.hpp"
#include "../../stable_diffusion_burn_amd/csrc/tokenizer.hpp"

namespace sdmi {
void write_png_rgb8(const std::string& path, const uint8_t* rgb, int width, int height);
}

static std::vector<unsigned char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& p, const unsigned char* d, size_t n) {
    std::ofstream f(p, std::ios::binary);
    f.write(reinterpret_cast<const char*>(d), (std::streamsize)n);
}

int main(int argc, char** argv) {
    if (argc != 4) { std::fprintf(stderr, "usage: host_san <merges> <record.mpk> <scratch dir>\n"); return 2; }
    const std::string merges = argv[1], record = argv[2], dir = argv[3];
    long checks = 0;

    {   // ---- tokenizer -------------------------------------------------------------------------------------------
        sdmi::Tokenizer tok(merges);
        std::vector<std::string> texts = {"", " ", "a photo of an astronaut riding a horse on mars", "HELLO   world!!  it's 2024's", "\xff\xfe\xfd",
                                          "caf\xc3\xa9 \xe2\x82\xac \xf0\x9f\x98\x80", std::string(5000, 'x'), "\xc3", "\xe2\x82", "tab\tnew\nline\r"};
        std::string all;
        for (int b = 1; b < 256; ++b) all.push_back((char)b);
        texts.push_back(all);
        std::mt19937 rng(1);
        for (int i = 0; i < 200; ++i) {
            std::string s;
            const int n = (int)(rng() % 60);
            for (int j = 0; j < n; ++j) s.push_back((char)(rng() % 255 + 1));
            texts.push_back(s);
        }
        for (const auto& t : texts) {
            const std::vector<int32_t> ids = tok.encode(t);
            const std::string back = tok.decode(ids.data(), ids.size());
            (void)back;
            ++checks;
        }
        try { const int32_t bad[2] = {-5, 1 << 30}; (void)tok.decode(bad, 2); } catch (const sdmi::Error&) {}
        ++checks;
    }
    {   // ---- PNG writer ------------------------------------------------------------------------------------------
        for (int wh : {1, 3, 64, 151}) {   // 151 x 151 x 3 + filter bytes > 65535: more than one stored deflate block
            std::vector<uint8_t> img((size_t)wh * wh * 3);
            for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)(i * 7);
            sdmi::write_png_rgb8(dir + "/t" + std::to_string(wh) + ".png", img.data(), wh, wh);
            ++checks;
        }
        try { sdmi::write_png_rgb8(dir + "/no/such/dir/x.png", nullptr, 0, 0); } catch (const sdmi::Error&) {}
    }
    {   // ---- .mpk reader ----------------------------------------------------------------------------------------
        const std::vector<unsigned char> good = slurp(record);
        if (good.empty()) { std::fprintf(stderr, "cannot read %s\n", record.c_str()); return 2; }
        size_t n_tensors = 0;
        {
            sdmi::MpkFile f(record);
            n_tensors = f.tensors().size();
            for (const auto& t : f.tensors()) {   // touch every tensor byte: the pointers must lie inside the mapping
                unsigned acc = 0;
                for (size_t i = 0; i < t.count * 4; ++i) acc += t.data[i];
                (void)acc;
            }
        }
        if (n_tensors == 0) return 3;
        const std::string tmp = dir + "/fuzz.mpk";
        long parsed = 0, rejected = 0;
        for (size_t cut = 0; cut < good.size(); cut += (good.size() > 4000 ? 7 : 1)) {
            spit(tmp, good.data(), cut);
            try { sdmi::MpkFile f(tmp); ++parsed; } catch (const sdmi::Error&) { ++rejected; }
            ++checks;
        }
        std::mt19937 rng(2);
        for (int i = 0; i < 3000; ++i) {
            std::vector<unsigned char> bad = good;
            const int flips = 1 + (int)(rng() % 3);
            for (int j = 0; j < flips; ++j) bad[rng() % bad.size()] = (unsigned char)rng();
            spit(tmp, bad.data(), bad.size());
            try {
                sdmi::MpkFile f(tmp);
                for (const auto& t : f.tensors()) { volatile unsigned char c = t.count ? t.data[t.count * 4 - 1] : 0; (void)c; }
                ++parsed;
            } catch (const sdmi::Error&) { ++rejected; } catch (const std::exception&) { ++rejected; }
            ++checks;
        }
        std::printf("mpk: %zu tensors; %ld mutated records parsed, %ld rejected\n", n_tensors, parsed, rejected);
    }
    std::printf("host_san: %ld checks, no sanitizer report\n", checks);
    return 0;
}
