// The fixed pixel sample of a frame used by the mean-field compensation: kSampleRun consecutive pixels
// from each of up to kSampleCells equal cells, every run at a hashed offset inside its cell.  Shared by norms.hip (sampled means)
// and rowchain.hip (the sampled pass of the fused block tail).
#pragma once

constexpr int kSampleRun = 16, kSampleCells = 64;
// cells_max: fewer, larger cells (a sparser sample) - pgt_frame_bias on bands of a frame keeps the number of sampled pixels per
// IMAGE bounded (16 bands x 16 cells x 16 pixels = 4096) instead of reading every pixel of small maps
__host__ __device__ inline void mean_sample_geometry(int HW, int* run, int* cells, int* cell, int cells_max = kSampleCells) {
    *run = HW < kSampleRun ? HW : kSampleRun;
    int r = HW / *run;
    if (cells_max < 1 || cells_max > kSampleCells) cells_max = kSampleCells;
    *cells = r < cells_max ? r : cells_max;
    *cell = HW / *cells;
}
// i-th sampled pixel: pixel i % run of the run of cell i / run, which starts at a hashed offset inside the cell (so that the
// runs do not line up in columns of the image)
__host__ __device__ inline int mean_sample_pixel(int i, int cell, int run) {
    const int j = i / run;
    return j * cell + (int)(((unsigned)j * 40503u) % (unsigned)(cell - run + 1)) + i % run;
}
