// formats_roundtrip.cpp — drives the host library's on-disk formats (print_diagnostic = the reference's
// impulse.dump, reference rayverb/helpers.cpp:19-59; write_ir_dump / read_ir_dump) for tests/test_formats.py.
//   formats_roundtrip dump <in.bin> <nrays> <nrefl> <out.dump>     raw Impulse records -> impulse.dump
//   formats_roundtrip ir <in.rvbh> <out.rvbh>                      read + re-write a binary IR dump
#include "rayverb/helpers.h"
#include "rayverb/ir_dump.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

int main(int argc, char ** argv)
{
    try {
        if (argc == 6 && !std::strcmp(argv[1], "dump")) {
            const unsigned long nrays = std::strtoul(argv[3], nullptr, 10), nrefl = std::strtoul(argv[4], nullptr, 10);
            std::vector<Impulse> imp(nrays * nrefl);
            std::ifstream in(argv[2], std::ios::binary);
            in.read(reinterpret_cast<char *>(imp.data()), (std::streamsize) (imp.size() * sizeof(Impulse)));
            if (!in) { std::cerr << "short input\n"; return 2; }
            print_diagnostic(nrays, nrefl, imp, argv[5]);
            return 0;
        }
        if (argc == 4 && !std::strcmp(argv[1], "ir")) {
            write_ir_dump(argv[3], read_ir_dump(argv[2]));
            return 0;
        }
    } catch (const std::exception & e) {
        std::cerr << "error: " << e.what() << "\n";
        return 3;
    }
    std::cerr << "usage\n";
    return 1;
}
