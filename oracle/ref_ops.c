/* ORACLE -- test infrastructure, NOT product code.
 *
 * Torch-free restatement (plain C loops, double accumulation) of the three operators of the SketchEdit
 * inference path, so the torch-based oracle (oracle/sketchedit_oracle.py) does not rest on
 * torch.nn.functional alone.  Small cases only.  Checked against the reference-generated golden vectors
 * in tests/test_oracle_golden.py.  All tensors fp32 NCHW, contiguous.
 *
 *   ref_gated_conv      /root/reference/models/networks/utils.py:9-51  (gen_conv, gen_deconv)
 *   ref_attention       /root/reference/models/networks/splitcam.py:37-108,132-153 (+ editline_g.py:204)
 */
#include <math.h>
#include <stdlib.h>

static double elu(double x) { return x > 0 ? x : expm1(x); }
static double sigm(double x) { return 1.0 / (1.0 + exp(-x)); }

/* act: 0 ELU gate, 1 ReLU gate, 2 none.  upsample: nearest x2 before a stride-1 conv.
 * y: (B, Cout/2, Ho, Wo) when gated, (B, Cout, Ho, Wo) when act==2 or Cout==3 (utils.py:27). */
void ref_gated_conv(const float* x, const float* w, const float* b, float* y, int B, int Cin, int H, int W, int Cout,
                    int k, int stride, int rate, int act, int upsample) {
  const int pad = rate * (k - 1) / 2;
  const int Hu = upsample ? 2 * H : H, Wu = upsample ? 2 * W : W;
  const int Ho = (Hu + 2 * pad - rate * (k - 1) - 1) / stride + 1, Wo = (Wu + 2 * pad - rate * (k - 1) - 1) / stride + 1;
  const int raw = (act == 2) || Cout == 3;
  double* pre = (double*)malloc(sizeof(double) * Cout);
  for (int n = 0; n < B; ++n)
    for (int oy = 0; oy < Ho; ++oy)
      for (int ox = 0; ox < Wo; ++ox) {
        for (int oc = 0; oc < Cout; ++oc) {
          double acc = b[oc];
          for (int ic = 0; ic < Cin; ++ic)
            for (int ky = 0; ky < k; ++ky)
              for (int kx = 0; kx < k; ++kx) {
                int iy = oy * stride + ky * rate - pad, ix = ox * stride + kx * rate - pad;
                if (iy < 0 || iy >= Hu || ix < 0 || ix >= Wu) continue;       /* zero padding */
                if (upsample) { iy >>= 1; ix >>= 1; }                          /* out[i,j] = in[i/2, j/2] */
                acc += (double)w[((oc * Cin + ic) * k + ky) * k + kx] * (double)x[((n * Cin + ic) * H + iy) * W + ix];
              }
          pre[oc] = acc;
        }
        if (raw) {
          for (int oc = 0; oc < Cout; ++oc) y[((n * Cout + oc) * Ho + oy) * Wo + ox] = (float)pre[oc];
        } else {
          const int G = Cout / 2;
          for (int c = 0; c < G; ++c) {
            const double f = act == 0 ? elu(pre[c]) : (pre[c] > 0 ? pre[c] : 0);
            y[((n * G + c) * Ho + oy) * Wo + ox] = (float)(f * sigm(pre[G + c]));
          }
        }
      }
  free(pre);
}

/* x (B,C,h,w); mask_full (B,1,4h,4w) hole mask; out (B,C,h,w); similar (B,L,hs,ws) softmax scores (may be NULL).
 * patch 4, stride 2, threshold 0.1, softmax scale 10. */
void ref_attention(const float* x, const float* mask_full, float* out, float* similar, int B, int C, int h, int w) {
  const int hs = (h - 4) / 2 + 1, ws = (w - 4) / 2 + 1, L = hs * ws, H = 4 * h, W = 4 * w;
  double* rn = (double*)malloc(sizeof(double) * C);
  double* S = (double*)malloc(sizeof(double) * L * L);     /* S[j*L + i], key j, query i */
  double* valid = (double*)malloc(sizeof(double) * L);
  for (int n = 0; n < B; ++n) {
    const float* xb = x + (size_t)n * C * h * w;
    for (int c = 0; c < C; ++c) {
      double ss = 0;
      for (int p = 0; p < h * w; ++p) ss += (double)xb[c * h * w + p] * xb[c * h * w + p];
      rn[c] = 1.0 / sqrt(ss + 1e-8);
    }
    for (int j = 0; j < L; ++j) {
      const int jy = j / ws, jx = j % ws;
      double hole = 0;   /* mean over the 4x4 patch of (1 - avg_pool2d(mask,4,4)) */
      for (int yy = 0; yy < 16; ++yy)
        for (int xx = 0; xx < 16; ++xx) hole += mask_full[((size_t)n * H + jy * 8 + yy) * W + jx * 8 + xx];
      valid[j] = (1.0 - hole / 256.0) > 0.1 ? 1.0 : 0.0;
    }
    for (int j = 0; j < L; ++j)
      for (int i = 0; i < L; ++i) {
        const int jy = j / ws, jx = j % ws, iy = i / ws, ix = i % ws;
        double acc = 0;
        for (int c = 0; c < C; ++c)
          for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx)
              acc += (double)xb[(c * h + 2 * jy + ky) * w + 2 * jx + kx] * rn[c] *
                     (double)xb[(c * h + 2 * iy + ky) * w + 2 * ix + kx];
        S[(size_t)j * L + i] = acc * valid[j] * 10.0;
      }
    for (int i = 0; i < L; ++i) {          /* softmax over keys */
      double m = -1e300, sum = 0;
      for (int j = 0; j < L; ++j) m = S[(size_t)j * L + i] > m ? S[(size_t)j * L + i] : m;
      for (int j = 0; j < L; ++j) { S[(size_t)j * L + i] = exp(S[(size_t)j * L + i] - m); sum += S[(size_t)j * L + i]; }
      for (int j = 0; j < L; ++j) S[(size_t)j * L + i] /= sum;
    }
    if (similar)
      for (size_t t = 0; t < (size_t)L * L; ++t) similar[(size_t)n * L * L + t] = (float)S[t];
    float* ob = out + (size_t)n * C * h * w;
    for (int c = 0; c < C; ++c)
      for (int y = 0; y < h; ++y)
        for (int xx = 0; xx < w; ++xx) {
          double acc = 0;   /* overlap-add of every patch covering (y,xx), no normalisation */
          for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx) {
              const int y2 = y - ky, x2 = xx - kx;
              if (y2 < 0 || x2 < 0 || (y2 & 1) || (x2 & 1)) continue;
              const int iy = y2 / 2, ix = x2 / 2;
              if (iy >= hs || ix >= ws) continue;
              const int i = iy * ws + ix;
              for (int j = 0; j < L; ++j)
                acc += S[(size_t)j * L + i] * (double)xb[(c * h + 2 * (j / ws) + ky) * w + 2 * (j % ws) + kx];
            }
          ob[(c * h + y) * w + xx] = (float)acc;
        }
  }
  free(rn); free(S); free(valid);
}
