// Padding contract between the weight packers and every kernel that streams packed weights with read-ahead.
//
// A packed tensor is [chunk][n tile][fragments of the chunk ...]: a kernel walks one n tile's fragments chunk by chunk
// and keeps `ahead` fragment loads in flight, so behind the last chunk it CONSUMES it still touches
// pad_chunks(ahead, steps) chunks (steps = fragments per chunk and n tile).  The packers append k*PadChunks zero chunks;
// every kernel instantiation static_asserts that its own read-ahead fits, and nd_conv*_max_weight_read() evaluates the
// same expressions per variant on the host (tests/test_host_logic.py checks them against nd_conv*_weight_{floats,elems}
// over a shape grid, split-K included).  History: a 16x16x32 bf16 1x1 stream (2 fragments per chunk, ring of 4) once ran
// one fragment past a single padding chunk and aborted the process -- the padding was a comment, not a contract.
#pragma once

namespace nd {
namespace wstream {

constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }
// zero chunks needed behind the last consumed chunk
constexpr int pad_chunks(int ahead, int steps) { return ceil_div(ahead, steps); }

// ---- fp32 direct / 1x1 / GEMM forms (nd_conv_weight_floats): chunk = 32 channels, fragment = 256 floats,
//      taps * 4 fragments per chunk and n tile
constexpr int kF32PadChunks = 1;
constexpr int f32_conv_ahead(int taps) { return taps == 9 ? 1 : 3; }      // conv_mfma_kernel's BDIST
constexpr int kF32StreamAhead = 3;                                         // gemm_stream_kernel's D
constexpr int kF32GemmAhead = 0;                                           // gemm_f32_kernel: stages are only issued for real chunks
constexpr int kF32Gemm4Ahead = 1;                                          // gemm4_kernel: one k-step

// ---- Winograd forms (nd_conv_winograd_weight_floats): chunk = 32 channels, 4 k-steps of 16 positions per chunk and n tile
constexpr int kWinoPadChunks = 1;
constexpr int kWinoStepsPerChunk = 4;
constexpr int kWinoAhead = 1;             // conv_wino_kernel, conv_wino16(p)_kernel, conv_wino4_kernel: one k-step
constexpr int kWinoDmaAhead = 2;          // conv_wino16g_kernel: two k-steps
constexpr int kWinoWaveAhead = 1;         // conv_winow_kernel: one half-step, never more than one k-step

// ---- Winograd F(4x4,3x3) form (nd_conv_winograd_f4_weight_floats): chunk = 16 channels, 4 k4-steps of 6 positions per chunk,
//      n block and transform row
constexpr int kWf4PadChunks = 1;
constexpr int kWf4StepsPerChunk = 4;
constexpr int kWf4Ahead = 1;              // conv_wf4_kernel: one k4-step

// ---- bf16 forms (nd_conv_bf16_weight_elems): chunk = 64 channels; taps * 4 fragments (32x32x16 layout) or taps * 2
//      fragments per 16-channel n tile (16x16x32 layout) per chunk
constexpr int kBf16PadChunks = 2;
constexpr int bf16_ring(int taps) { return taps == 9 ? 3 : 4; }                                   // conv_bf16_kernel
constexpr int bf16s_ring(int tn, int taps) { return tn >= 4 ? 2 : (taps == 9 ? 3 : 4); }          // conv_bf16s_kernel
constexpr int bf16_steps(int taps) { return taps * 4; }
constexpr int bf16s_steps(int taps) { return taps * 2; }
constexpr int kBf16GemmQAheadSteps = 4; // gemm_bf16q_kernel: the next chunk's four k-steps
constexpr int kBf16DmaAheadTaps = 1;      // conv_bf16w_kernel: the next tap's stage (a chunk has `taps` of them)

}  // namespace wstream
}  // namespace nd
