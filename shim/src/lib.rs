//! datafusion-gpu: DataFusion 36 `ExecutionPlan` twins over `libdfgpu.so` (MI355X / gfx950).
//!
//! * [`ffi`]      the C ABI of `include/dfgpu.h` + `include/dfgpu_exec.h`, RAII handles, status -> `DataFusionError`
//! * [`subplan`]  `GpuSubplanExec`: a maximal GPU-supported subtree of the physical plan as ONE `ExecutionPlan`
//!                (`execute` = `dfgpu_plan_execute`, `poll_next` = `dfgpu_stream_next` + export)
//! * [`rule`]     `GpuOffload`: the `PhysicalOptimizerRule` that substitutes such subtrees
//!
//! Registration (core/src/execution/context/mod.rs:1563-1569):
//! ```ignore
//! let state = SessionState::new_with_config_rt(config, runtime)
//!     .add_physical_optimizer_rule(Arc::new(datafusion_gpu::rule::GpuOffload::default()));
//! ```
//! Nothing inside `datafusion/physical-plan` changes; a node the device library answers `NotImplemented` for keeps its CPU operator.
pub mod ffi;
pub mod rule;
pub mod subplan;
