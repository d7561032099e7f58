// reader_fuzz_main.cpp — drives the native dump and log readers of libmdhip.so's host side over files given on the
// command line (tests/test_reader_asan_cpu.py builds it together with csrc/dump_reader.cpp under
// -fsanitize=address,undefined). Every frame / run is read completely; failures must be error returns, never faults.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/mdhip.h"

int main(int argc, char **argv)
{
    long frames = 0, runs = 0, errors = 0;
    for (int k = 1; k < argc; ++k) {
        mdhip_dump *d = nullptr;
        if (mdhip_dump_open(argv[k], &d) == 0 && d) {
            const int64_t nf = mdhip_dump_n_frames(d);
            for (int64_t f = 0; f < nf; ++f) {
                int64_t ts = 0, na = 0;
                double b6[6], t3[3];
                int tri = 0, nc = 0;
                char names[4096];
                if (mdhip_dump_frame_info(d, f, &ts, &na, b6, t3, &tri, &nc, names, sizeof names) != 0) {
                    ++errors;
                    continue;
                }
                if (nc <= 0 || na < 0 || na > 10000000) continue;
                std::vector<int32_t> idx(nc);
                for (int c = 0; c < nc; ++c) idx[c] = c;
                std::vector<double> out((size_t)nc * (size_t)(na > 0 ? na : 1));
                for (int sort_col = -1; sort_col < (nc > 0 ? 1 : 0); ++sort_col)
                    for (int threads = 1; threads <= 3; threads += 2)
                        if (mdhip_dump_read(d, f, nc, idx.data(), sort_col, out.data(), threads) != 0) ++errors;
                ++frames;
            }
            mdhip_dump_close(d);
        } else {
            ++errors;
        }
        mdhip_log *l = nullptr;
        if (mdhip_log_open(argv[k], &l) == 0 && l) {
            const int64_t nr = mdhip_log_n_runs(l);
            for (int64_t r = 0; r < nr; ++r) {
                int64_t rows = 0;
                int cols = 0, regular = 0;
                char names[4096];
                if (mdhip_log_run_info(l, r, &rows, &cols, &regular, names, sizeof names) != 0) {
                    ++errors;
                    continue;
                }
                std::vector<double> out((size_t)(cols > 0 ? cols : 1) * (size_t)(rows > 0 ? rows : 1));
                std::vector<int32_t> is_int(cols > 0 ? cols : 1);
                for (int threads = 1; threads <= 3; threads += 2)
                    if (mdhip_log_read(l, r, out.data(), is_int.data(), threads) != 0) ++errors;
                ++runs;
            }
            mdhip_log_close(l);
        } else {
            ++errors;
        }
    }
    printf("frames %ld runs %ld error-returns %ld\n", frames, runs, errors);
    return 0;
}
