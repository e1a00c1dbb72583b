/* CPU oracle, C part.  TEST INFRASTRUCTURE ONLY (see mtam_oracle.py header).
 *
 * score_fma: scores[m][n] = sum_k A[m][k] * Bt[n][k] accumulated as a k-ordered
 * float32 fmaf chain starting from 0 -- the arithmetic of
 * tf.matmul(pred, item_table^T) (Model/base_model.py:195,316) with the
 * accumulation order fixed to the one the gfx950 fp32 MFMA implements, so that
 * scores (and therefore tf.nn.top_k rankings, Model/base_model.py:196-200) can be
 * compared bit for bit.  TF itself does not promise an accumulation order.
 */
#include <math.h>
#include <stddef.h>

void score_fma(const float *A, const float *Bt, float *C, int M, int N, int K) {
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float acc = 0.0f;
      const float *a = A + (size_t)m * K, *b = Bt + (size_t)n * K;
      for (int k = 0; k < K; ++k) acc = fmaf(a[k], b[k], acc);
      C[(size_t)m * N + n] = acc;
    }
}
