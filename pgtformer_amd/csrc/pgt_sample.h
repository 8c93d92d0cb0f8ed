// The fixed pixel sample of a frame used by the mean-field compensation: kSampleRun consecutive pixels
// from each of up to kSampleCells equal cells, every run at a hashed offset inside its cell.  Shared by norms.hip (sampled means)
// and rowchain.hip (the sampled pass of the fused block tail).
#pragma once

constexpr int kSampleRun = 16, kSampleCells = 64;
__host__ __device__ inline void mean_sample_geometry(int HW, int* run, int* cells, int* cell) {
    *run = HW < kSampleRun ? HW : kSampleRun;
    int r = HW / *run;
    *cells = r < kSampleCells ? r : kSampleCells;
    *cell = HW / *cells;
}
// i-th sampled pixel: pixel i % run of the run of cell i / run, which starts at a hashed offset inside the cell (so that the
// runs do not line up in columns of the image)
__host__ __device__ inline int mean_sample_pixel(int i, int cell, int run) {
    const int j = i / run;
    return j * cell + (int)(((unsigned)j * 40503u) % (unsigned)(cell - run + 1)) + i % run;
}
