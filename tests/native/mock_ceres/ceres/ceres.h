// Minimal stand-in for the three Ceres interfaces lvi-exc_amd/host/lvx_ceres_shim.hpp builds on — TEST INFRASTRUCTURE ONLY (the image has no
// Ceres).  Signatures follow Ceres' public headers (ceres/cost_function.h, ceres/evaluation_callback.h); nothing else of Ceres is modelled.
#pragma once
#include <cstdint>
#include <vector>
namespace ceres {
class CostFunction {
 public:
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int32_t>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }
 protected:
  std::vector<int32_t>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }
 private:
  std::vector<int32_t> parameter_block_sizes_;
  int num_residuals_ = 0;
};
class EvaluationCallback {
 public:
  virtual ~EvaluationCallback() {}
  virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};
}  // namespace ceres
