// ref_conv_driver.cc -- runs the REFERENCE's own CPU Conv / ConvGradient operators.
//
// TEST INFRASTRUCTURE ONLY; built only in the container that has /root/reference (oracle/Makefile
// target `refconv`, recipe oracle/build_ref_conv.sh).  Everything that computes here is the
// reference's code compiled from where it lies: ConvOp<float, CPUContext> and
// ConvGradientOp<float, CPUContext> (caffe2/caffe2/operators/conv_op_impl.h:31-202, :358-577) with
// math::Im2col / Col2im / Gemm / Gemv of caffe2/caffe2/utils/math_cpu.cc (Eigen backend,
// CAFFE2_USE_EIGEN_FOR_BLAS -- the reference's own build option), created through the
// reference's operator registry from an OperatorDef exactly as its Python layer does.
// This file only feeds blobs and fetches results.
#include <cstring>
#include <string>
#include <vector>

#include "caffe2/core/operator.h"
#include "caffe2/core/workspace.h"

namespace {

void feed(caffe2::Workspace* ws, const std::string& name, const float* data, const std::vector<caffe2::TIndex>& dims) {
  auto* t = ws->CreateBlob(name)->GetMutable<caffe2::TensorCPU>();
  t->Resize(dims);
  std::memcpy(t->mutable_data<float>(), data, t->size() * sizeof(float));
}

void fetch(caffe2::Workspace* ws, const std::string& name, float* out, size_t expect) {
  const auto& t = ws->GetBlob(name)->Get<caffe2::TensorCPU>();
  CAFFE_ENFORCE_EQ((size_t)t.size(), expect, "unexpected size of ", name);
  std::memcpy(out, t.data<float>(), expect * sizeof(float));
}

void add_int(caffe2::OperatorDef* def, const char* name, int v) {
  auto* a = def->add_arg();
  a->set_name(name);
  a->set_i(v);
}

void conv_args(caffe2::OperatorDef* def, int kernel, int pad, int stride, int group) {
  add_int(def, "kernel", kernel);
  add_int(def, "pad", pad);
  add_int(def, "stride", stride);
  if (group != 1) add_int(def, "group", group);
  auto* a = def->add_arg();
  a->set_name("order");
  a->set_s("NCHW");
}

int out_size(int in, int kernel, int pad, int stride) { return (in + 2 * pad - kernel) / stride + 1; }

}  // namespace

extern "C" {

// Y[N][M][OH][OW] = Conv(X[N][C][H][W], filter[M][C/group][k][k], bias[M] or NULL)
__attribute__((visibility("default"))) int ref_conv_forward(
    const float* X, const float* filter, const float* bias, int N, int C, int H, int W, int M, int kernel,
    int pad, int stride, int group, float* Y) {
  try {
    caffe2::Workspace ws;
    feed(&ws, "X", X, {N, C, H, W});
    feed(&ws, "w", filter, {M, C / group, kernel, kernel});
    caffe2::OperatorDef def;
    def.set_type("Conv");
    def.add_input("X");
    def.add_input("w");
    if (bias) {
      feed(&ws, "b", bias, {M});
      def.add_input("b");
    }
    def.add_output("Y");
    conv_args(&def, kernel, pad, stride, group);
    auto op = caffe2::CreateOperator(def, &ws);
    if (!op || !op->Run()) return 2;
    fetch(&ws, "Y", Y, (size_t)N * M * out_size(H, kernel, pad, stride) * out_size(W, kernel, pad, stride));
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "ref_conv_forward: %s\n", e.what());
    return 1;
  }
}

// ConvGradient [X, filter, dY] -> [dfilter, dbias, dX] (caffe2/operators/conv_gradient_op.cc:35-77)
__attribute__((visibility("default"))) int ref_conv_backward(
    const float* X, const float* filter, const float* dY, int N, int C, int H, int W, int M, int kernel,
    int pad, int stride, int group, float* dfilter, float* dbias, float* dX) {
  try {
    caffe2::Workspace ws;
    const int OH = out_size(H, kernel, pad, stride), OW = out_size(W, kernel, pad, stride);
    feed(&ws, "X", X, {N, C, H, W});
    feed(&ws, "w", filter, {M, C / group, kernel, kernel});
    feed(&ws, "dY", dY, {N, M, OH, OW});
    caffe2::OperatorDef def;
    def.set_type("ConvGradient");
    def.add_input("X");
    def.add_input("w");
    def.add_input("dY");
    def.add_output("dw");
    def.add_output("db");
    def.add_output("dX");
    conv_args(&def, kernel, pad, stride, group);
    auto op = caffe2::CreateOperator(def, &ws);
    if (!op || !op->Run()) return 2;
    fetch(&ws, "dw", dfilter, (size_t)M * (C / group) * kernel * kernel);
    fetch(&ws, "db", dbias, (size_t)M);
    fetch(&ws, "dX", dX, (size_t)N * C * H * W);
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "ref_conv_backward: %s\n", e.what());
    return 1;
  }
}

}  // extern "C"
