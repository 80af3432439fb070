// ONNX-subset graph runtime: the engine-side replacement for the three onnxruntime sessions InsightFace's
// FaceAnalysis opens for the reference (analyzers/face.py:30-38: det_10g / 2d106det / w600k_r50 of buffalo_l).
// The model files are the reference's own weights format, so the engine reads them directly: a dependency-free
// protobuf wire reader, then an interpreter that runs every node on the HIP kernels of this library in NHWC with
// Conv+BatchNorm+activation+residual chains fused into one launch. Nothing here links protobuf / onnx / onnxruntime.
#pragma once
#include "engine.h"
#include "onnx_model.h"
#include <deque>
#include <set>

namespace fe {

struct GraphOutput {
  std::string name;
  std::vector<int64_t> dims;   // ONNX logical layout (NCHW for feature maps)
  float* dev = nullptr;        // dense fp32, arena memory (valid until the arena is reset)
  size_t numel = 0;
};

class Graph {
 public:
  void load(const uint8_t* data, size_t len);
  // x: NHWC view whose first `lc` channels are the logical input channels (the rest are zero padding).
  // Launches on c.stream, allocates from c.arena; outs follow the model's declared output order.
  void run(Ctx& c, const Tensor& x, int lc, std::vector<GraphOutput>& outs);
  const onnx::Model& model() const { return m_; }
  // InsightFace decides the input normalisation of landmark / recognition models by looking for Sub / Mul among the
  // first 8 node names [DEP-KNOWLEDGE insightface model_zoo]; the host mirror needs the same two bits.
  bool head_has_sub() const { return has_sub_; }
  bool head_has_mul() const { return has_mul_; }
  size_t weight_bytes() const { return dw_.bytes(); }
  static int pad_channels(int c) { return c <= 20 ? ((c + 3) & ~3) : ((c + 15) & ~15); }

 private:
  struct Val {
    enum Kind { NONE, IMG, PLAIN, HOST } kind = NONE;
    Tensor t;                          // IMG: NHWC physical view (c = padded channels)
    int lc = 0;                        // IMG: logical channels
    int rank = 4;                      // IMG: 4, or 2 when h == w == 1 stands for an [N, C] matrix
    bool flat = false;                 // IMG rank 4 seen through Flatten(axis=1): logical [N, C*H*W] in NCHW order
    float* p = nullptr;                // PLAIN: dense row-major
    std::vector<int64_t> dims;         // PLAIN
    const onnx::TensorData* host = nullptr;   // HOST: initializer / constant / shape arithmetic result
  };
  struct Group { int main = -1, bn = -1, act = -1, add = -1, act2 = -1; };
  struct NodeCache {
    bool built = false;
    ConvW cw;
    float* dwt = nullptr;              // depthwise weights [taps][C]
    float* scale = nullptr; float* shift = nullptr; float* slope = nullptr;
    long long key = 0;                 // layout the pack was made for (input channels / flatten geometry)
  };

  onnx::Model m_;
  DeviceWeights dw_;
  std::vector<Group> group_end_;       // indexed by node; main >= 0 where a fused group executes
  std::vector<char> absorbed_;
  std::vector<NodeCache> cache_;
  std::map<std::string, int> uses_;
  std::map<std::string, int> producer_;
  bool has_sub_ = false, has_mul_ = false;

  // per-run state
  std::map<std::string, Val> vals_;
  std::deque<onnx::TensorData> temps_;

  const Val& get(const std::string& name);
  const onnx::TensorData* host_of(const std::string& name);   // null when not a host value
  bool is_const(const std::string& name) const;
  void plan_fusion();
  void exec_group(Ctx& c, int idx);
  void exec_node(Ctx& c, int idx);
  Val to_img(Ctx& c, const Val& v);
  Val to_plain(Ctx& c, const Val& v);
  void set_host(const std::string& name, onnx::TensorData&& t);
  void channel_vector(const onnx::TensorData& t, int C, int Cp, float padv, std::vector<float>& out) const;
};

// One loaded graph plus the host copies of its last outputs (fe_graph_run / fe_graph_output_*).
struct GraphSlot {
  Graph g;
  struct Out { std::string name; std::vector<int64_t> dims; std::vector<float> data; };
  std::vector<Out> last;
};

}  // namespace fe
