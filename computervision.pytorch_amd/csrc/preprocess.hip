// Input side of the detection path on gfx950: the reference's letter_box (core/utils/image_process.py:48-66) + TF.to_tensor
// (:41) in one pass -- uint8 HWC image in device memory -> one (3, H, W) fp32 slot of the network's input batch.
//
//   scale = min(H / h, W / w);  new_h, new_w = int(h * scale), int(w * scale)          (Python floats = C doubles, truncation)
//   cv2.resize(image, (new_w, new_h), INTER_NEAREST):  src x = min(floor(x * (1 / (new_w / w))), w - 1)   (OpenCV resizeNN;
//                                                      the reciprocal of the forward scale, both in double, like cv::resize)
//   copyMakeBorder(top = (H - new_h) // 2, left = (W - new_w) // 2, value 128)
//   to_tensor: HWC uint8 -> CHW float32 / 255                                                  (a true fp32 division)
// Byte and index work: bit-exact against the restatement in oracle/letterbox_ref.py.  HBM-bound: one thread per output pixel
// writes its three planes (coalesced along x), reads three bytes.
#include "cvx_common.h"
#include "../../include/cvx_engine.h"

namespace {

__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* src, int h, int w, int new_h, int new_w, int top, int left, double ify,
                                                        double ifx, int swap_rb, float pad, float* dst, int H, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const int yy = y - top, xx = x - left;
  float v0 = pad, v1 = pad, v2 = pad;
  if (yy >= 0 && yy < new_h && xx >= 0 && xx < new_w) {
    int sy = (int)floor((double)yy * ify), sx = (int)floor((double)xx * ifx);
    sy = sy < h - 1 ? sy : h - 1;
    sx = sx < w - 1 ? sx : w - 1;
    const uint8_t* p = src + ((long long)sy * w + sx) * 3;
    v0 = (float)p[swap_rb ? 2 : 0] / 255.0f;
    v1 = (float)p[1] / 255.0f;
    v2 = (float)p[swap_rb ? 0 : 2] / 255.0f;
  }
  dst[i] = v0;
  dst[(long long)H * W + i] = v1;
  dst[2LL * H * W + i] = v2;
}

}  // namespace

extern "C" int cvx_letterbox_geometry(int32_t h, int32_t w, int32_t H, int32_t W, int32_t* new_h, int32_t* new_w, int32_t* top, int32_t* left,
                                      double* scale) {
  CVX_CHECK(h > 0 && w > 0 && H > 0 && W > 0 && new_h && new_w && top && left && scale, "bad arguments");
  const double sh = (double)H / (double)h, sw = (double)W / (double)w;
  const double s = sh < sw ? sh : sw;
  *scale = s;
  *new_h = (int32_t)((double)h * s);
  *new_w = (int32_t)((double)w * s);
  CVX_CHECK(*new_h > 0 && *new_w > 0, "letterbox: the image collapses to nothing at this size");
  *top = (H - *new_h) / 2;
  *left = (W - *new_w) / 2;
  return 0;
}

extern "C" int cvx_letterbox_u8_to_nchw(const uint8_t* image_hwc, int32_t h, int32_t w, int32_t letterbox, int32_t swap_rb, float* out_chw,
                                        int32_t H, int32_t W, void* hip_stream) {
  CVX_CHECK(image_hwc && out_chw && h > 0 && w > 0 && H > 0 && W > 0, "bad arguments");
  int32_t nh = H, nw = W, top = 0, left = 0;
  double scale = 0;
  if (letterbox) CVX_TRY(cvx_letterbox_geometry(h, w, H, W, &nh, &nw, &top, &left, &scale));
  // cv::resize: inv_scale = dsize / ssize, scale = 1 / inv_scale (doubles); resizeNN takes floor(dst * scale)
  const double ify = 1.0 / ((double)nh / (double)h), ifx = 1.0 / ((double)nw / (double)w);
  hipLaunchKernelGGL(letterbox_kernel, dim3(cvx_cdiv((long long)H * W, 256)), dim3(256), 0, (hipStream_t)hip_stream, image_hwc, h, w, nh, nw, top,
                     left, ify, ifx, swap_rb, 128.0f / 255.0f, out_chw, H, W);
  CVX_HIP(hipGetLastError());
  return 0;
}
