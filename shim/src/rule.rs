//! `GpuOffload`: the `PhysicalOptimizerRule` (core/src/physical_optimizer/optimizer.rs:39-55) that puts `GpuSubplanExec` in place of
//! GPU-supported subtrees -- the pattern of the built-in `CoalesceBatches` rule (core/src/physical_optimizer/coalesce_batches.rs:40-94).
use std::sync::Arc;

use datafusion::physical_optimizer::PhysicalOptimizerRule;
use datafusion_common::config::ConfigOptions;
use datafusion_common::tree_node::{Transformed, TreeNode};
use datafusion_common::Result;
use datafusion_physical_plan::ExecutionPlan;

use crate::subplan::{gpu_type, is_gpu_node, GpuSubplanExec};

#[derive(Debug)]
pub struct GpuOffload {
    /// HIP device the plan segments run on
    pub device: i32,
    /// smallest estimated input (rows) worth a trip over PCIe; segments below it keep their CPU operators
    pub min_rows: usize,
}
impl Default for GpuOffload { fn default() -> Self { Self { device: 0, min_rows: 1 << 20 } } }

/// Every column that crosses the segment's boundary must have a device type.
fn schema_supported(p: &dyn ExecutionPlan) -> bool { p.schema().fields().iter().all(|f| gpu_type(f.data_type()).is_some()) }

/// Does the subtree do work the device is good at (a join, an aggregation or a sort), not only filters and projections over a CPU scan?
fn has_heavy_operator(p: &Arc<dyn ExecutionPlan>) -> bool {
    let n = p.name();
    n == "HashJoinExec" || n == "AggregateExec" || n == "SortExec" || p.children().iter().any(|c| is_gpu_node(c.as_ref()) && has_heavy_operator(c))
}

impl PhysicalOptimizerRule for GpuOffload {
    fn optimize(&self, plan: Arc<dyn ExecutionPlan>, _config: &ConfigOptions) -> Result<Arc<dyn ExecutionPlan>> {
        // top-down: the first (highest) device node of a branch takes its whole device-supported subtree; what is below a
        // GpuSubplanExec is not visited again (its CPU leaves are executed by the subplan itself)
        plan.transform_down(&|node: Arc<dyn ExecutionPlan>| {
            if node.as_any().is::<GpuSubplanExec>() { return Ok(Transformed::No(node)); }
            let rows = node.statistics().ok().and_then(|s| s.num_rows.get_value().copied()).unwrap_or(usize::MAX);
            if is_gpu_node(node.as_ref()) && schema_supported(node.as_ref()) && has_heavy_operator(&node) && rows >= self.min_rows {
                return Ok(Transformed::Yes(Arc::new(GpuSubplanExec::new(node, self.device))));
            }
            Ok(Transformed::No(node))
        })
    }
    fn name(&self) -> &str { "gpu_offload" }
    fn schema_check(&self) -> bool { true }
}
