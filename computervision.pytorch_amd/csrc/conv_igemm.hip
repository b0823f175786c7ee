// Convolution dispatcher: validates a ConvParams block and routes it to one of the three implicit-GEMM kernels
//   * conv_pw.hip         1x1 stride-1 forward / data gradient: persistent, barrier-free, weights resident in LDS;
//   * conv_halo.hip       3x3 stride-1 forward / data gradient: persistent, halo patch + weights resident in LDS;
//   * conv_gemm.hip       big-channel layers (taps * Cin >= 1024, Cin % 32 == 0): 256 x 256 ... 128 x 128 macro tiles, both operands
//                         through an LDS-DMA ring, v_mfma_f32_32x32x16_f16;
//   * conv_igemm_dma.hip  every other shape (stride 2, 256+ channels, dilation, merged stride-2 gradient phases):
//                         LDS-DMA ring.
// (The register-staged first-generation kernel that used to live here was retired once the DMA kernels covered every
// shape; the dispatcher keeps the file name.)
#include "conv_igemm.h"

unsigned long long* g_cvx_clk = nullptr;
int g_cvx_grid_div = 1;

int cvx_conv_igemm_launch(const ConvParams& p_in, hipStream_t stream, int* m_blocks) {
  ConvParams p = p_in;
  p.clk = g_cvx_clk;
  CVX_CHECK(p.Cin % 8 == 0 && p.in_ld % 8 == 0 && p.wt_ld % 8 == 0, "conv: channel counts must be multiples of 8");
  CVX_CHECK(p.Cout % 4 == 0 && p.out_ld % 4 == 0, "conv: Cout/out_ld must be multiples of 4");
  CVX_CHECK(p.ntaps >= 1 && p.ntaps <= CVX_MAX_TAPS, "conv: bad tap count");
  CVX_CHECK(((uintptr_t)p.in % 16) == 0 && ((uintptr_t)p.wt % 16) == 0, "conv: operands must be 16-byte aligned");
  CVX_CHECK(p.zeros && ((uintptr_t)p.zeros % 16) == 0, "conv: needs a 16-byte aligned zero page (padding source of the LDS DMA)");
  CVX_CHECK((long long)p.B * p.OH2 * p.OW2 > 0, "conv: empty output");
  if (m_blocks) *m_blocks = 0;
  if (p.ps_cin > 0) {  // stride-2 data gradient as one GEMM with a pixel-shuffle store: the GEMM-shaped kernel's plain epilogue only
    CVX_CHECK(p.epi == CVX_EPI_PLAIN && p.nphase <= 1 && p.OS == 2 && p.oph == 0 && p.opw == 0 && p.ps_cin % 8 == 0 && p.Cout == 4 * p.ps_cin &&
                  cvx_conv_gemm_shape_ok(p),
              "conv: pixel-shuffle data gradient needs the GEMM-shaped kernel (plain epilogue, 4 phases of ps_cin channels, output stride 2)");
    return cvx_conv_gemm_launch(p, stream);
  }
  if (p.nphase > 1) return cvx_conv_igemm_dma_launch(p, stream);  // merged phases: the DMA-ring kernel only
  if (cvx_conv_stem7_supported(p)) return cvx_conv_stem7_launch(p, stream);  // 7x7 first layer on the padded image
  if (cvx_conv_tile_supported(p)) return cvx_conv_tile_launch(p, stream);  // small 3x3 stride-1 maps: one row band per workgroup
  if (cvx_conv_gemm_supported(p)) return cvx_conv_gemm_launch(p, stream);  // big-channel layers, whatever their taps
  if (cvx_conv_pw_supported(p)) return cvx_conv_pw_launch(p, stream);
  if (cvx_conv_halo_supported(p)) return cvx_conv_halo_launch(p, stream);
  return cvx_conv_igemm_dma_launch(p, stream);
}
