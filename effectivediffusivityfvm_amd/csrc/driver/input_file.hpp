// input_file.hpp -- the reference's input.txt, parsed compatibly.
//
// Format (reference readInputFile, Deff2DGPU/Deff2D.cuh:234-324; doc section 3): one
// `Key: value` pair per line, the key INCLUDING its colon is the first whitespace-delimited
// token, case-sensitive, any order; numeric values are read as double and cast ("MaxIter: 5e5"
// works, cuh:299-300); the three file names are the second token of their line; unknown lines
// (e.g. the decorative first line "Input File:") are ignored.  The reference leaves unset
// fields uninitialised; here every field has a default and the result is validated.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

namespace deff {

struct Options {                   // same meaning as the reference's `options`, cuh:18-37
    int nPhase = 2;
    double DCsolid = 0.0, DCfluid = 1.0, DCgas = 1.0;
    int MeshIncreaseX = 1, MeshIncreaseY = 1;
    double CLeft = 0.0, CRight = 1.0;
    long MAX_ITER = 500000;
    double ConvergeCriteria = 1e-6;
    std::string inputFilename = "00000.jpg", outputFilename = "output.csv", CMapName = "CMAP.csv";
    int printCmap = 0, verbose = 0, BatchFlag = 0, NumImg = 1;
};

// double -> integer as the reference's casts do for sane values; NaN and out-of-range values
// (undefined behaviour in a plain cast) are pinned to 0 / the range ends and fail validation later
inline long to_long(double v)
{
    if (!(v == v)) return 0;
    if (v >= 9.0e18) return (long)9000000000000000000L;
    if (v <= -9.0e18) return (long)-9000000000000000000L;
    return (long)v;
}
inline int to_int(double v)
{
    const long l = to_long(v);
    return l > 2147483647L ? 2147483647 : (l < -2147483647L ? -2147483647 : (int)l);
}

inline bool read_input_file(const char *path, Options *o, std::string *err)
{
    std::ifstream in(path);
    if (!in) { *err = std::string("cannot open ") + path; return false; }
    std::string line;
    while (std::getline(in, line)) {
        char key[1000] = "", text[1000] = "";
        double v = 0;
        const int got = std::sscanf(line.c_str(), "%999s %lf", key, &v);
        if (got < 1) continue;
        std::sscanf(line.c_str(), "%*s %999s", text);
        const std::string k = key;
        if (k == "Ds:") o->DCsolid = v;
        else if (k == "Df:") o->DCfluid = v;
        else if (k == "Dg:") o->DCgas = v;
        else if (k == "MeshAmpX:") o->MeshIncreaseX = to_int(v);
        else if (k == "MeshAmpY:") o->MeshIncreaseY = to_int(v);
        else if (k == "InputName:") o->inputFilename = text;
        else if (k == "CR:") o->CRight = v;
        else if (k == "CL:") o->CLeft = v;
        else if (k == "OutputName:") o->outputFilename = text;
        else if (k == "printCMap:") o->printCmap = to_int(v);
        else if (k == "CMapName:") o->CMapName = text;
        else if (k == "Convergence:") o->ConvergeCriteria = v;
        else if (k == "MaxIter:") o->MAX_ITER = to_long(v);
        else if (k == "Verbose:") o->verbose = to_int(v);
        else if (k == "RunBatch:") o->BatchFlag = to_int(v);
        else if (k == "NumImages:") o->NumImg = to_int(v);
        else if (k == "Phases:") o->nPhase = to_int(v);
    }
    if (o->nPhase != 2 && o->nPhase != 3) { *err = "Phases must be 2 or 3"; return false; }
    if (o->MeshIncreaseX < 1 || o->MeshIncreaseY < 1) {          // cuh:1901-1904
        *err = "MeshIncrease has to be an integer greater than 1.";
        return false;
    }
    if (o->verbose != 0 && o->verbose != 1)
        std::printf("Please enter a value of 0 or 1 for 'verbose'. Default = 0.\n");   // cuh:320-322
    if (o->MAX_ITER < 1) { *err = "MaxIter must be >= 1"; return false; }
    if (!(o->ConvergeCriteria == o->ConvergeCriteria)) { *err = "Convergence is not a number"; return false; }
    if (o->BatchFlag && o->NumImg < 1) { *err = "NumImages must be >= 1 in batch mode"; return false; }
    return true;
}

// Options echo under Verbose: 1 (what the reference's printOptions reports, cuh:121-175).
inline void print_options(const Options &o)
{
    std::printf("--------------------------------------\n");
    std::printf("Effective diffusivity (FVM), MI355X-native solver\n");
    std::printf("Phases = %d\n", o.nPhase);
    std::printf("Ds = %g, Df = %g", o.DCsolid, o.DCfluid);
    if (o.nPhase == 3) std::printf(", Dg = %g", o.DCgas);
    std::printf("\nMesh amplification = %d x %d\n", o.MeshIncreaseX, o.MeshIncreaseY);
    std::printf("CL = %g, CR = %g\n", o.CLeft, o.CRight);
    std::printf("Convergence = %g, MaxIter = %ld\n", o.ConvergeCriteria, o.MAX_ITER);
    if (o.BatchFlag) std::printf("Batch of %d images (%%05d.jpg)\n", o.NumImg);
    else std::printf("Input = %s\n", o.inputFilename.c_str());
    std::printf("Output = %s%s%s\n", o.outputFilename.c_str(), o.printCmap ? ", CMap = " : "",
                o.printCmap ? o.CMapName.c_str() : "");
    std::printf("--------------------------------------\n");
}

}  // namespace deff
