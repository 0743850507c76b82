/*
 * TEST INFRASTRUCTURE — host program side of the reference's C ABI.
 *
 * The reference controller library (template/uprightmpc2/uprightmpc2.c) does
 * not define `matMult`; it IMPORTS it from whatever host links it
 * (template/uprightmpc2/matmult.h:22-35 "C = alpha * op(A) * op(B)", column
 * major; template/uprightmpc2/uprightmpc2.h:17 "The matMult function must be
 * defined somewhere - dependent on C++ or C").  Every real host of the
 * reference (pybind module, Simulink S-function, MCU firmware) supplies its
 * own.  This file is OUR host: it supplies that callback with a plain triple
 * loop.  It contains no reference code and replaces no reference header,
 * library or generated file.
 */
void matMult(float *C, const float *A, const float *B, const int m,
             const int n, const int k, const float alpha, int AT, int BT) {
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < m; ++i) {
      float acc = 0.0f;
      for (int l = 0; l < k; ++l) {
        const float a = AT ? A[l + i * k] : A[i + l * m];
        const float b = BT ? B[j + l * n] : B[l + j * k];
        acc += a * b;
      }
      C[i + j * m] = alpha * acc;
    }
  }
}
