// frontend_fuzz.cpp -- robustness of the host-side front end on damaged input (CPU only).
// Feeds truncated, bit-flipped and spliced variants of a good JPEG to the decoder and random
// key/value soup to the input-file parser.  Built with -fsanitize=address,undefined: any
// out-of-bounds access, overflow or leak aborts the run.  Usage: frontend_fuzz good.jpg rounds
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include "driver/input_file.hpp"
#include "driver/jpeg_gray.hpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s good.jpg rounds\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> good((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const int rounds = atoi(argv[2]);
    if (good.size() < 100) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }

    std::vector<uint8_t> pix;
    std::string err;
    int w, h, n;
    if (!deff::jpeg::decode_gray(good.data(), good.size(), pix, w, h, n, err)) { fprintf(stderr, "good file rejected: %s\n", err.c_str()); return 1; }
    const size_t good_px = pix.size();
    long accepted = 0, rejected = 0;

    for (int r = 0; r < rounds; ++r) {
        std::vector<uint8_t> v = good;
        switch (rnd() % 6) {
        case 0: v.resize(rnd() % v.size()); break;                                   // truncate anywhere
        case 1: for (int k = 0, m = 1 + (int)(rnd() % 8); k < m; ++k) v[rnd() % v.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 2: for (int k = 0, m = 1 + (int)(rnd() % 64); k < m; ++k) v[rnd() % v.size()] = (uint8_t)rnd(); break;
        case 3: {                                                                    // damage the headers only
            const size_t lim = v.size() < 700 ? v.size() : 700;
            for (int k = 0, m = 1 + (int)(rnd() % 6); k < m; ++k) v[rnd() % lim] = (uint8_t)rnd();
            break;
        }
        case 4: {                                                                    // splice a chunk elsewhere
            const size_t a = rnd() % v.size(), b = rnd() % v.size(), len = rnd() % 512;
            for (size_t k = 0; k < len && a + k < v.size() && b + k < v.size(); ++k) v[a + k] = v[b + k];
            break;
        }
        default: {                                                                   // SOI + garbage
            v.resize(2 + rnd() % 4096);
            for (size_t k = 2; k < v.size(); ++k) v[k] = (uint8_t)rnd();
            break;
        }
        }
        std::vector<uint8_t> out;
        std::string e;
        int ww = 0, hh = 0, nn = 0;
        bool ok = false;
        try {
            ok = deff::jpeg::decode_gray(v.data(), v.size(), out, ww, hh, nn, e);
        } catch (const std::exception &ex) {                                         // e.g. bad_alloc on an absurd size
            e = ex.what();
        }
        if (ok) {
            if (out.size() != (size_t)ww * hh || ww <= 0 || hh <= 0) { fprintf(stderr, "accepted with inconsistent size\n"); return 1; }
            ++accepted;
        } else {
            if (e.empty()) { fprintf(stderr, "rejected without a message\n"); return 1; }
            ++rejected;
        }
    }

    // input.txt parser: random soup of known keys, junk values, missing colons, long lines
    static const char *keys[] = {"Phases", "Ds", "Df", "Dg", "MeshAmpX", "MeshAmpY", "InputName", "CR", "CL", "OutputName",
                                 "printCMap", "CMapName", "Convergence", "MaxIter", "RunBatch", "NumImages", "Nope", ""};
    static const char *vals[] = {"1", "0", "-3", "1e-3", "1e400", "nan", "abc", "", "  ", "2 3", "0x10", "99999999999999999999", ":", "a:b"};
    long parsed = 0;
    for (int r = 0; r < rounds; ++r) {
        std::string text;
        for (int k = 0, m = (int)(rnd() % 24); k < m; ++k) {
            text += keys[rnd() % (sizeof keys / sizeof *keys)];
            if (rnd() % 8) text += (rnd() % 4) ? ": " : ":";
            text += vals[rnd() % (sizeof vals / sizeof *vals)];
            if (rnd() % 16 == 0) text += std::string(rnd() % 5000, 'x');
            text += (rnd() % 10) ? "\n" : "\r\n";
        }
        const std::string path = std::string(argv[1]) + ".fuzz_input.txt";
        { std::ofstream o(path, std::ios::binary); o << text; }
        deff::Options in;
        std::string e;
        if (deff::read_input_file(path.c_str(), &in, &e)) ++parsed;
        else if (e.empty()) { fprintf(stderr, "input rejected without a message\n"); return 1; }
        remove(path.c_str());
    }
    printf("jpeg: %ld accepted, %ld rejected (good image %zu px); input files: %ld of %d parsed\n", accepted, rejected, good_px,
           parsed, rounds);
    return 0;
}
