// exec.cpp -- C++ host layer mirroring the reference's operator interface for the hot path.
//
//   trait PhysicalExpr   datafusion/physical-expr/src/physical_expr.rs:96-123
//   trait ExecutionPlan  datafusion/physical-plan/src/lib.rs:115-405
//   MemoryExec memory.rs | FilterExec filter.rs:315-363 | ProjectionExec projection.rs:295-340 |
//   CoalesceBatchesExec coalesce_batches.rs:198-260 | CoalescePartitionsExec | RepartitionExec repartition/mod.rs:148-294 |
//   HashJoinExec joins/hash_join.rs:574-1388 | AggregateExec aggregates/{mod,row_hash}.rs | SortExec sorts/sort.rs:584-988
//
// It only uses the public kernel-level C ABI (include/dfgpu.h): exactly what a Rust shim would do per operator.
// MI355X-first differences from the reference (results identical): batches are whole partitions; FilterExec emits a
// selection bitmap that join build/probe, group interning and repartitioning consume fused; join outputs are lazy
// (source, indices) columns so that columns a later ProjectionExec drops are never gathered.
#include "../../../include/dfgpu_exec.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <atomic>
#include <deque>
#include <map>
#include <tuple>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace dfx {

struct Err : std::exception {
  dfgpu_status code; std::string msg;
  Err(dfgpu_status c, std::string m) : code(c), msg(std::move(m)) {}
  const char* what() const noexcept override { return msg.c_str(); }
};
[[noreturn]] static void fail(dfgpu_status code, const char* fmt, ...) {
  char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  throw Err(code, buf);
}
static thread_local std::string g_err;

// ------------------------------------------------------------------ handles
struct ArrayRef {
  dfgpu_array* a = nullptr;
  ArrayRef() = default;
  static ArrayRef adopt(dfgpu_array* owned) { ArrayRef r; r.a = owned; return r; }
  static ArrayRef share(const dfgpu_array* x) { ArrayRef r; r.a = const_cast<dfgpu_array*>(x); if (r.a) dfgpu_array_retain(r.a); return r; }
  ArrayRef(const ArrayRef& o) : a(o.a) { if (a) dfgpu_array_retain(a); }
  ArrayRef(ArrayRef&& o) noexcept : a(o.a) { o.a = nullptr; }
  ArrayRef& operator=(ArrayRef o) { std::swap(a, o.a); return *this; }
  ~ArrayRef() { if (a) dfgpu_array_release(a); }
  explicit operator bool() const { return a != nullptr; }
  int64_t len() const { return a ? dfgpu_array_length(a) : 0; }
};

struct TaskContext {          // ≙ datafusion_execution::TaskContext (execution/src/task.rs:44-59)
  dfgpu_ctx* ctx; int64_t batch_size; bool metrics = false;       // metrics: option "collect_metrics" when the stream was created
  void check(dfgpu_status st) const { if (st != DFGPU_OK) throw Err(st, dfgpu_last_error(ctx)); }
};

struct Field { std::string name; int32_t type = 0, precision = 0, scale = 0; };
struct Schema { std::vector<Field> f; };
using SchemaPtr = std::shared_ptr<Schema>;

static Field field_of(const std::string& name, const dfgpu_array* a) {
  dfgpu_array_desc d; dfgpu_array_describe(a, &d);
  if (d.type == DFGPU_DICTIONARY && d.dictionary) return Field{name, d.dictionary->type, d.dictionary->precision, d.dictionary->scale};
  return Field{name, d.type, d.precision, d.scale};
}

// a column is a materialised array or a pending gather (late materialisation).  The gather's row i reads
// source[chain[0][chain[1][...[i]]]]: gathering a pending column only appends to the chain, and the chain collapses from its short
// (outer) end when the column is finally read, so an operator that keeps few of a huge join output's rows (TPC-H Q18's semi join)
// never composes index arrays at the join output's length.
struct Composed { ArrayRef inner, outer, out; };
// Columns that came out of one operator share their chain arrays; they also share this memo so each composition runs once, not once
// per column.  Entries pin the arrays their key points at.
struct TakeMemo { std::map<std::pair<const dfgpu_array*, const dfgpu_array*>, Composed> composed; };
using MemoPtr = std::shared_ptr<TakeMemo>;
static ArrayRef take(const TaskContext& tc, const ArrayRef& v, const ArrayRef& idx) { dfgpu_array* o = nullptr; tc.check(dfgpu_take(tc.ctx, v.a, idx.a, &o)); return ArrayRef::adopt(o); }
// The innermost stage of a pending gather can be a join's build-row lookup that has not run yet (dfgpu_join_probe_deferred): the chain then indexes the join's matched
// pairs, and the build rows are computed when a build-side column is finally read -- for the pairs still wanted by then.  TPC-H Q18 joins 600 M lineitems to their orders
// and keeps a few thousand of them in the next (semi) join: the lookup runs for those, not for 600 M.  Columns of one join output share the object, so a lookup runs once
// per distinct index array.
struct LazyLookup {
  std::shared_ptr<void> keep; dfgpu_join_table* table = nullptr; ArrayRef probe_key, rows; int64_t m = 0;      // rows: the matched probe rows (UInt32), m of them
  std::mutex mu; std::map<const dfgpu_array*, std::pair<ArrayRef, ArrayRef>> done;                              // pair-index array (pinned) -> build rows; nullptr = every pair
  ArrayRef resolve(const TaskContext& tc, const ArrayRef& idx) {
    std::lock_guard<std::mutex> l(mu);
    auto it = done.find(idx.a); if (it != done.end()) return it->second.second;
    if (idx) { auto full = done.find(nullptr); if (full != done.end()) { ArrayRef out = take(tc, full->second.second, idx); done[idx.a] = std::make_pair(idx, out); return out; } }      // every pair is known already: a gather, not a second lookup
    ArrayRef sel = rows;
    if (idx) {
      ArrayRef i32 = idx; dfgpu_array_desc d; dfgpu_array_describe(idx.a, &d);
      if (d.type != DFGPU_UINT32) { dfgpu_array* c = nullptr; tc.check(dfgpu_cast(tc.ctx, idx.a, DFGPU_UINT32, 0, 0, &c)); i32 = ArrayRef::adopt(c); }
      sel = dfgpu_array_is_identity(rows.a) ? i32 : take(tc, rows, i32);
    }
    const dfgpu_array* kp = probe_key.a; dfgpu_array* b = nullptr; tc.check(dfgpu_join_lookup(tc.ctx, table, &kp, 1, sel.a, &b)); ArrayRef out = ArrayRef::adopt(b);
    done[idx.a] = std::make_pair(idx, out); return out;
  }
};
struct Col {
  ArrayRef arr, source; std::vector<ArrayRef> chain; MemoPtr memo; std::shared_ptr<LazyLookup> lookup;
  int64_t len() const { return arr ? arr.len() : !chain.empty() ? chain.back().len() : lookup ? lookup->m : 0; }
};
static const ArrayRef& col_indices(const TaskContext& tc, Col& c) {           // the chain as one index array into c.source
  while (c.chain.size() > 1) {
    ArrayRef outer = std::move(c.chain.back()); c.chain.pop_back();
    ArrayRef& inner = c.chain.back();
    if (c.memo) {
      auto key = std::make_pair((const dfgpu_array*)inner.a, (const dfgpu_array*)outer.a);
      auto it = c.memo->composed.find(key);
      if (it == c.memo->composed.end()) it = c.memo->composed.emplace(key, Composed{inner, outer, take(tc, inner, outer)}).first;
      inner = it->second.out;
    } else inner = take(tc, inner, outer);
  }
  if (c.lookup) { ArrayRef b = c.lookup->resolve(tc, c.chain.empty() ? ArrayRef() : c.chain[0]); c.chain.clear(); c.chain.push_back(b); c.lookup.reset(); }
  return c.chain[0];
}
static const ArrayRef& col_get(const TaskContext& tc, Col& c) {
  if (!c.arr) { c.arr = take(tc, c.source, col_indices(tc, c)); c.source = ArrayRef(); c.chain.clear(); c.memo.reset(); }
  return c.arr;
}
static Col col_take(const Col& c, const ArrayRef& idx, const MemoPtr& memo = nullptr) {
  if (dfgpu_array_is_identity(idx.a) && idx.len() == c.len()) return c;          // every row, in order: the column itself
  Col o; o.memo = memo;
  if (c.arr) { o.source = c.arr; o.chain.push_back(idx); return o; }
  o.source = c.source; o.chain = c.chain; o.lookup = c.lookup; o.chain.push_back(idx);
  return o;
}
static Col col_of(ArrayRef a) { Col c; c.arr = std::move(a); return c; }

struct Batch {
  SchemaPtr schema; std::vector<Col> cols; ArrayRef selection; int64_t base_rows = 0;
  // A pending gather of a result-sized batch (<= 2^20 rows) takes its siblings along: every other column of the batch that waits behind the SAME index arrays is gathered in the
  // same launch (dfgpu_take_multi) -- at that size a launch costs more than the bytes it moves, and a query's last operators read 3-5 columns through each row list.
  const ArrayRef& column(const TaskContext& tc, int i) {
    Col& c = cols.at((size_t)i);
    if (!c.arr && c.source && !c.chain.empty() && base_rows <= ((int64_t)1 << 20) && base_rows > 0) {
      std::vector<size_t> sib;
      for (size_t k = 0; k < cols.size(); k++) { const Col& o = cols[k]; if (k == (size_t)i || o.arr || !o.source || o.lookup.get() != c.lookup.get() || o.chain.size() != c.chain.size()) continue;
        bool same = true; for (size_t q = 0; q < c.chain.size(); q++) same = same && o.chain[q].a == c.chain[q].a; if (same) sib.push_back(k); }
      if (!sib.empty()) {
        const ArrayRef idx = col_indices(tc, c);               // composes the chain / resolves a deferred lookup once
        std::vector<const dfgpu_array*> vals{ c.source.a }; for (size_t k : sib) vals.push_back(cols[k].source.a);
        std::vector<dfgpu_array*> outs(vals.size(), nullptr);
        tc.check(dfgpu_take_multi(tc.ctx, vals.data(), (int32_t)vals.size(), idx.a, outs.data()));
        auto settle = [](Col& x, dfgpu_array* a) { x.arr = ArrayRef::adopt(a); x.source = ArrayRef(); x.chain.clear(); x.memo.reset(); x.lookup.reset(); };
        settle(c, outs[0]); for (size_t u = 0; u < sib.size(); u++) settle(cols[sib[u]], outs[u + 1]);
        return c.arr;
      }
    }
    return col_get(tc, c);
  }
};
static ArrayRef mask_indices(const TaskContext& tc, const ArrayRef& mask) { dfgpu_array* o = nullptr; tc.check(dfgpu_mask_to_indices(tc.ctx, mask.a, &o)); return ArrayRef::adopt(o); }
// ≙ filter_record_batch (filter.rs:325), lazily per column
static Batch materialize(const TaskContext& tc, const Batch& b) {
  if (!b.selection) return b;
  ArrayRef sel = mask_indices(tc, b.selection);
  Batch o; o.schema = b.schema; o.base_rows = sel.len();
  MemoPtr memo = std::make_shared<TakeMemo>(); for (auto& c : b.cols) o.cols.push_back(col_take(c, sel, memo));
  return o;
}
static int64_t num_rows(const TaskContext& tc, const Batch& b) { return b.selection ? mask_indices(tc, b.selection).len() : b.base_rows; }
static ArrayRef concat_arrays(const TaskContext& tc, std::vector<ArrayRef>& parts) {
  std::vector<const dfgpu_array*> p; for (auto& x : parts) p.push_back(x.a);
  dfgpu_array* o = nullptr; tc.check(dfgpu_concat(tc.ctx, p.data(), (int32_t)p.size(), &o)); return ArrayRef::adopt(o);
}
// concat_batches (arrow-select), after applying selections; empty batches are skipped
static bool concat_batches(const TaskContext& tc, std::vector<Batch>& in, Batch* out) {
  std::vector<Batch> m;
  for (auto& b : in) { Batch x = materialize(tc, b); if (x.base_rows > 0) m.push_back(std::move(x)); }
  if (m.empty()) { if (in.empty()) return false; *out = materialize(tc, in[0]); return true; }
  if (m.size() == 1) { *out = std::move(m[0]); return true; }
  Batch o; o.schema = m[0].schema;
  for (size_t c = 0; c < m[0].cols.size(); c++) {
    std::vector<ArrayRef> parts; for (auto& b : m) parts.push_back(col_get(tc, b.cols[c]));
    o.cols.push_back(col_of(concat_arrays(tc, parts)));
  }
  for (auto& b : m) o.base_rows += b.base_rows;
  *out = std::move(o); return true;
}

// ------------------------------------------------------------------ PhysicalExpr
struct Value { ArrayRef arr; bool scalar = false; };      // ≙ ColumnarValue (expr/src/columnar_value.rs:35-40)
struct Expr {
  virtual ~Expr() = default;
  virtual Value eval(const TaskContext& tc, Batch& b) const = 0;
  virtual bool safe() const { return false; }            // cannot raise on rows a selection mask has dropped
  virtual void columns(std::set<int>& out) const {}
  virtual int column_index() const { return -1; }
};
using ExprPtr = std::shared_ptr<const Expr>;

static ArrayRef into_array(const TaskContext& tc, const Value& v, int64_t n) {       // ColumnarValue::into_array
  if (!v.scalar) return v.arr;
  dfgpu_array* idx = nullptr; tc.check(dfgpu_array_new_zeros(tc.ctx, DFGPU_UINT32, 0, 0, n, &idx));     // broadcast = take(scalar, [0; n])
  ArrayRef i = ArrayRef::adopt(idx);
  return take(tc, v.arr, i);
}

struct ColumnExpr : Expr {        // expressions/column.rs:91
  std::string name; int index;
  ColumnExpr(std::string n, int i) : name(std::move(n)), index(i) {}
  Value eval(const TaskContext& tc, Batch& b) const override {
    if (index < 0 || index >= (int)b.cols.size())
      fail(DFGPU_INTERNAL, "PhysicalExpr Column references column '%s' at index %d (zero-based) but input schema only has %zu columns", name.c_str(), index, b.cols.size());
    return Value{b.column(tc, index), false};
  }
  bool safe() const override { return true; }
  void columns(std::set<int>& out) const override { out.insert(index); }
  int column_index() const override { return index; }
};
struct LiteralExpr : Expr {       // expressions/literal.rs:73
  ArrayRef scalar;
  explicit LiteralExpr(ArrayRef s) : scalar(std::move(s)) {}
  Value eval(const TaskContext&, Batch&) const override { return Value{scalar, true}; }
  bool safe() const override { return true; }
};
struct BinaryExpr : Expr {        // expressions/binary.rs:259-315
  ExprPtr l, r; int op;
  BinaryExpr(ExprPtr a, int o, ExprPtr b) : l(std::move(a)), r(std::move(b)), op(o) {}
  static bool arith_op(int o) { return o >= DFGPU_OP_ADD && o <= DFGPU_OP_REM; }
  // `x op (literal op2 y)` / `(literal op2 y) op x`: one fused pass when the device takes the shape (dfgpu_binary_fused2), else node by node
  bool eval_fused(const TaskContext& tc, Batch& b, Value* out) const {
    if (!arith_op(op)) return false;
    for (int inner_left = 0; inner_left < 2; inner_left++) {
      auto* in = dynamic_cast<const BinaryExpr*>((inner_left ? l : r).get());
      if (!in || !arith_op(in->op)) continue;
      auto* lit_l = dynamic_cast<const LiteralExpr*>(in->l.get()); auto* lit_r = dynamic_cast<const LiteralExpr*>(in->r.get());
      if ((lit_l != nullptr) == (lit_r != nullptr)) continue;                     // exactly one literal side
      Value x = (inner_left ? r : l)->eval(tc, b), y = (lit_l ? in->r : in->l)->eval(tc, b);
      if (x.scalar || y.scalar) return false;
      dfgpu_array* o = nullptr;
      dfgpu_status st = dfgpu_binary_fused2(tc.ctx, op, x.arr.a, in->op, (lit_l ? lit_l : lit_r)->scalar.a, y.arr.a, lit_l ? 1 : 0, inner_left, &o);
      if (st == DFGPU_NOT_IMPLEMENTED) {           // node by node, with the children already evaluated
        const ArrayRef& lit = (lit_l ? lit_l : lit_r)->scalar; dfgpu_array* t = nullptr;
        tc.check(lit_l ? dfgpu_binary(tc.ctx, in->op, lit.a, 1, y.arr.a, 0, &t) : dfgpu_binary(tc.ctx, in->op, y.arr.a, 0, lit.a, 1, &t));
        ArrayRef inner = ArrayRef::adopt(t);
        tc.check(inner_left ? dfgpu_binary(tc.ctx, op, inner.a, 0, x.arr.a, 0, &o) : dfgpu_binary(tc.ctx, op, x.arr.a, 0, inner.a, 0, &o));
      } else tc.check(st);
      *out = Value{ArrayRef::adopt(o), false}; return true;
    }
    return false;
  }
  Value eval(const TaskContext& tc, Batch& b) const override {
    Value fused; if (eval_fused(tc, b, &fused)) return fused;
    Value x = l->eval(tc, b), y = r->eval(tc, b);
    dfgpu_array* o = nullptr; tc.check(dfgpu_binary(tc.ctx, op, x.arr.a, x.scalar, y.arr.a, y.scalar, &o));
    return Value{ArrayRef::adopt(o), x.scalar && y.scalar};
  }
  bool safe() const override { return op >= DFGPU_OP_EQ && l->safe() && r->safe(); }      // comparisons / AND / OR never raise
  void columns(std::set<int>& out) const override { l->columns(out); r->columns(out); }
};
struct UnaryExpr : Expr {
  ExprPtr e; int kind, a0, a1, a2; ArrayRef list;      // kind: 0 NOT, 1 IS NULL, 2 IS NOT NULL, 3 negative, 4 cast, 5 IN, 6 NOT IN
  Value eval(const TaskContext& tc, Batch& b) const override {
    Value x = e->eval(tc, b); dfgpu_array* o = nullptr;
    switch (kind) {
      case 0: tc.check(dfgpu_not(tc.ctx, x.arr.a, &o)); break;
      case 1: tc.check(dfgpu_is_null(tc.ctx, x.arr.a, 0, &o)); break;
      case 2: tc.check(dfgpu_is_null(tc.ctx, x.arr.a, 1, &o)); break;
      case 3: tc.check(dfgpu_negative(tc.ctx, x.arr.a, &o)); break;
      case 4: tc.check(dfgpu_cast(tc.ctx, x.arr.a, a0, a1, a2, &o)); break;
      default: tc.check(dfgpu_in_list(tc.ctx, x.arr.a, list.a, kind == 6, &o)); break;
    }
    return Value{ArrayRef::adopt(o), x.scalar};
  }
  bool safe() const override { return kind <= 2 && e->safe(); }
  void columns(std::set<int>& out) const override { e->columns(out); }
};

// ------------------------------------------------------------------ ExecutionPlan
struct Stream { virtual ~Stream() = default; virtual bool next(Batch& out) = 0; };     // poll_next: false = end of stream
struct Plan;
using PlanPtr = std::shared_ptr<const Plan>;
// ≙ BaselineMetrics (output_rows, elapsed_compute; physical-plan/src/metrics/baseline.rs:47-56), BuildProbeJoinMetrics (build_time, join_time;
// joins/utils.rs:1368-1426) and RepartitionMetrics (repartition_time; repartition/mod.rs:312-349).  Times are DEVICE times of the spans the
// operator enqueued (dfgpu_span_*), resolved when the metrics are read; elapsed_compute excludes the children's spans.
struct Metrics {
  std::mutex mu; int64_t output_rows = 0, output_batches = 0;
  int64_t spill_count = 0, spilled_bytes = 0, spilled_rows = 0;        // SortExec (sorts/sort.rs:229-231, 306-312)
  struct Open { dfgpu_ctx* c; int64_t id; int which; };        // which: 0 inclusive compute, 1 build, 2 join, 3 repartition
  std::vector<Open> open; int64_t ns[4] = {0, 0, 0, 0};
  void resolve() {
    std::lock_guard<std::mutex> l(mu);
    for (auto& o : open) { int64_t t = 0; if (dfgpu_span_elapsed_ns(o.c, o.id, &t) == DFGPU_OK) ns[o.which] += t; }
    open.clear();
  }
  ~Metrics() { resolve(); }          // a plan whose metrics nobody read gives its spans' events back (dfgpu_span_* refuse a ctx that is gone)
};
struct SpanGuard {
  dfgpu_ctx* c = nullptr; int64_t id = -1; Metrics* m = nullptr; int which = 0;
  SpanGuard(const TaskContext& tc, Metrics* met, int w) { if (tc.metrics && met && dfgpu_span_begin(tc.ctx, &id) == DFGPU_OK) { c = tc.ctx; m = met; which = w; } else id = -1; }
  ~SpanGuard() { if (id >= 0) { dfgpu_span_end(c, id); std::lock_guard<std::mutex> l(m->mu); m->open.push_back(Metrics::Open{c, id, which}); } }
};
struct Plan : std::enable_shared_from_this<Plan> {
  virtual ~Plan() = default;
  mutable std::shared_ptr<Metrics> met = std::make_shared<Metrics>();
  virtual std::vector<std::shared_ptr<const Plan>> children() const { return {}; }
  std::unique_ptr<Stream> run(int partition, const TaskContext& tc) const;          // execute() + metering
  virtual const char* name() const = 0;
  virtual SchemaPtr schema() const = 0;
  virtual int partitions() const = 0;                                   // output_partitioning().partition_count()
  virtual std::unique_ptr<Stream> execute(int partition, const TaskContext& tc) const = 0;
  // ≙ ExecutionPlan::with_new_children over the same (recursively fresh) children (physical-plan/src/lib.rs:198-201): a new node
  // without the run-once state of this one (HashJoinExec's OnceAsync build side, RepartitionExec's pulled input), so a plan
  // description built once can be executed again from scratch.
  virtual PlanPtr fresh() const = 0;
};
static int64_t metric_rows(const TaskContext& tc, const Batch& b) { if (!b.selection) return b.base_rows; int64_t k = 0; tc.check(dfgpu_mask_count(tc.ctx, b.selection.a, &k)); return k; }
struct MeteredStream : Stream {
  const Plan* op; PlanPtr keep; std::unique_ptr<Stream> in; TaskContext tc;
  MeteredStream(PlanPtr k, std::unique_ptr<Stream> i, TaskContext t) : op(k.get()), keep(std::move(k)), in(std::move(i)), tc(t) {}
  bool next(Batch& out) override {
    bool ok; { SpanGuard sp(tc, op->met.get(), 0); ok = in->next(out); }
    if (ok) { int64_t r = metric_rows(tc, out); std::lock_guard<std::mutex> l(op->met->mu); op->met->output_rows += r; op->met->output_batches++; }
    return ok;
  }
};
std::unique_ptr<Stream> Plan::run(int partition, const TaskContext& tc) const {
  if (!tc.metrics) return execute(partition, tc);
  std::unique_ptr<Stream> s; { SpanGuard sp(tc, met.get(), 0); s = execute(partition, tc); }       // operators that drain their input in execute()
  return std::unique_ptr<Stream>(new MeteredStream(shared_from_this(), std::move(s), tc));
}
static void drain(const PlanPtr& p, int partition, const TaskContext& tc, std::vector<Batch>& out) {
  auto s = p->run(partition, tc); Batch b; while (s->next(b)) out.push_back(std::move(b));
}

struct VecStream : Stream {
  std::vector<Batch> v; size_t i = 0;
  explicit VecStream(std::vector<Batch> x) : v(std::move(x)) {}
  bool next(Batch& out) override { if (i >= v.size()) return false; out = std::move(v[i++]); return true; }
};

struct MemoryExec : Plan {        // memory.rs:40,150
  mutable std::mutex mu; mutable std::vector<std::vector<Batch>> parts; SchemaPtr sch;      // parts: replaceable (dfgpu_plan_memory_replace), read under mu
  const char* name() const override { return "MemoryExec"; }
  PlanPtr fresh() const override { return shared_from_this(); }
  SchemaPtr schema() const override { return sch; }
  int partitions() const override { std::lock_guard<std::mutex> l(mu); return (int)parts.size(); }
  std::unique_ptr<Stream> execute(int p, const TaskContext&) const override {
    std::lock_guard<std::mutex> l(mu);
    if (p < 0 || p >= (int)parts.size()) fail(DFGPU_INTERNAL, "MemoryExec invalid partition %d (expected less than %zu)", p, parts.size());
    return std::unique_ptr<Stream>(new VecStream(parts[(size_t)p]));
  }
};

// ≙ ParquetExec (core/src/datasource/physical_plan/parquet/mod.rs:78-117, execute :356-414): one file, its row groups dealt to the output
// partitions in contiguous runs (the reference splits a file's byte range over target_partitions, file_groups / repartition_file_groups);
// a projection by leaf index; row groups whose statistics cannot satisfy a [min, max] bound on an integer / date column are skipped
// (≙ parquet/row_groups.rs prune_row_groups_by_statistics -- the FilterExec above still runs).  Batches are whole row groups (the kernels
// want large launches), not batch_size slices.
struct ParquetExec : Plan {
  dfgpu_parquet* file = nullptr; std::vector<int32_t> proj; SchemaPtr sch; int nparts = 1, per_batch = 1;
  struct Bound { int32_t col; int64_t lo, hi; }; mutable std::mutex mu; mutable std::vector<Bound> bounds; mutable std::atomic<int64_t> pruned{0};
  const char* name() const override { return "ParquetExec"; }
  PlanPtr fresh() const override { return shared_from_this(); }
  SchemaPtr schema() const override { return sch; }
  int partitions() const override { return nparts; }
  bool keep(int rg) const {
    std::lock_guard<std::mutex> l(mu);
    for (auto& b : bounds) { int64_t mn, mx, nc; int32_t has; if (dfgpu_parquet_column_stats(file, rg, b.col, &mn, &mx, &nc, &has) == DFGPU_OK && has && (mx < b.lo || mn > b.hi)) return false; }
    return true;
  }
  struct S : Stream {
    const ParquetExec* op; TaskContext tc; int next_rg, end_rg;
    S(const ParquetExec* o, TaskContext t, int a, int b) : op(o), tc(t), next_rg(a), end_rg(b) {}
    bool next(Batch& out) override {
      while (next_rg < end_rg && !op->keep(next_rg)) { next_rg++; op->pruned++; }
      if (next_rg >= end_rg) return false;
      int first = next_rg, n = 0;
      while (next_rg < end_rg && n < op->per_batch && op->keep(next_rg)) { next_rg++; n++; }
      std::vector<dfgpu_array*> cols(op->proj.size(), nullptr);
      tc.check(dfgpu_parquet_read(tc.ctx, op->file, first, n, op->proj.data(), (int32_t)op->proj.size(), cols.data()));
      Batch b; b.schema = op->sch; b.base_rows = 0; for (int g = first; g < first + n; g++) b.base_rows += dfgpu_parquet_row_group_rows(op->file, g);
      for (auto* a : cols) b.cols.push_back(col_of(ArrayRef::adopt(a)));
      out = std::move(b); return true;
    }
  };
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override {
    if (p < 0 || p >= nparts) fail(DFGPU_INTERNAL, "ParquetExec invalid partition %d (expected less than %d)", p, nparts);
    int64_t R = dfgpu_parquet_num_row_groups(file);
    return std::unique_ptr<Stream>(new S(this, tc, (int)(R * p / nparts), (int)(R * (p + 1) / nparts)));
  }
};

// ≙ CsvExec (core/src/datasource/physical_plan/csv.rs:53-72, execute :222-260, CsvOpener::open :327-420): one file image under the table's schema.  The reference
// deals byte ranges of the file to partitions and moves each range's ends to the next record boundary (find_first_newline, csv.rs:362-420); here the image is cut
// once, at plan time, into pieces of about `batch_bytes` that end on a record boundary outside quotes, the pieces are dealt to the partitions in contiguous runs and
// every piece is one batch parsed on the device (dfgpu_csv_read).  The caller keeps the image alive while the plan lives.
struct CsvExec : Plan {
  const uint8_t* bytes = nullptr; int64_t len = 0; int32_t delim = ',', quote = '"', escape = 0, ncols_file = 0; bool header = true;
  std::vector<int32_t> proj, types; SchemaPtr sch; int nparts = 1; std::vector<int64_t> cuts;      // piece i = [cuts[i], cuts[i + 1])
  const char* name() const override { return "CsvExec"; }
  PlanPtr fresh() const override { return shared_from_this(); }
  SchemaPtr schema() const override { return sch; }
  int partitions() const override { return nparts; }
  void cut(int64_t batch_bytes) {
    cuts.assign(1, 0); bool inside = false; int64_t next = batch_bytes;
    for (int64_t i = 0; i < len; i++) { const uint8_t c = bytes[i]; if (escape > 0 && c == (uint8_t)escape && inside && i + 1 < len) { i++; continue; } if (c == (uint8_t)quote) inside = !inside; else if (c == '\n' && !inside && i + 1 >= next && i + 1 < len) { cuts.push_back(i + 1); next = i + 1 + batch_bytes; } }
    cuts.push_back(len);
  }
  struct S : Stream {
    const CsvExec* op; TaskContext tc; int next_piece, end_piece;
    S(const CsvExec* o, TaskContext t, int a, int b) : op(o), tc(t), next_piece(a), end_piece(b) {}
    bool next(Batch& out) override {
      if (next_piece >= end_piece) return false;
      const int i = next_piece++; std::vector<dfgpu_array*> cols(op->proj.size(), nullptr); int64_t rows = 0;
      tc.check(dfgpu_csv_read(tc.ctx, op->bytes + op->cuts[(size_t)i], op->cuts[(size_t)i + 1] - op->cuts[(size_t)i], 0, op->delim, op->quote, op->escape, i == 0 && op->header ? 1 : 0, op->ncols_file,
                              op->proj.data(), op->types.data(), (int32_t)op->proj.size(), cols.data(), &rows));
      Batch b; b.schema = op->sch; b.base_rows = rows;
      for (auto* a : cols) b.cols.push_back(col_of(ArrayRef::adopt(a)));
      out = std::move(b); return true;
    }
  };
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override {
    if (p < 0 || p >= nparts) fail(DFGPU_INTERNAL, "CsvExec invalid partition %d (expected less than %d)", p, nparts);
    const int64_t P = (int64_t)cuts.size() - 1;
    return std::unique_ptr<Stream>(new S(this, tc, (int)(P * p / nparts), (int)(P * (p + 1) / nparts)));
  }
};

static ArrayRef known_mask(const TaskContext& tc, const ArrayRef& m) {     // NULL -> false before AND-ing selections
  dfgpu_array_desc d; dfgpu_array_describe(m.a, &d);
  if (!d.validity) return m;
  dfgpu_array *nn = nullptr, *o = nullptr;
  tc.check(dfgpu_is_null(tc.ctx, m.a, 1, &nn)); ArrayRef n1 = ArrayRef::adopt(nn);
  tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_AND, m.a, 0, n1.a, 0, &o)); return ArrayRef::adopt(o);
}

struct FilterExec : Plan {        // filter.rs:56-66, batch_filter :315-327
  ExprPtr pred; PlanPtr input;
  PlanPtr fresh() const override { auto f = std::make_shared<FilterExec>(); f->pred = pred; f->input = input->fresh(); return f; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {input}; }
  const char* name() const override { return "FilterExec"; }
  SchemaPtr schema() const override { return input->schema(); }
  int partitions() const override { return input->partitions(); }
  struct S : Stream {
    const FilterExec* op; std::unique_ptr<Stream> in; TaskContext tc;
    S(const FilterExec* o, std::unique_ptr<Stream> i, TaskContext t) : op(o), in(std::move(i)), tc(t) {}
    bool next(Batch& out) override {
      Batch b; if (!in->next(b)) return false;
      if (b.selection && !op->pred->safe()) b = materialize(tc, b);
      Value v = op->pred->eval(tc, b);
      ArrayRef mask = into_array(tc, v, b.base_rows);
      dfgpu_array_desc d; dfgpu_array_describe(mask.a, &d);
      if (d.type != DFGPU_BOOL) fail(DFGPU_INTERNAL, "Cannot create filter_array from non-boolean predicates");
      if (b.selection) { ArrayRef km = known_mask(tc, mask); dfgpu_array* o = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_AND, km.a, 0, b.selection.a, 0, &o)); mask = ArrayRef::adopt(o); }
      b.selection = mask; out = std::move(b); return true;
    }
  };
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override { return std::unique_ptr<Stream>(new S(this, input->run(p, tc), tc)); }
};

static Batch materialize_subset(const TaskContext& tc, Batch& b, const std::set<int>& needed) {   // compact only referenced columns
  ArrayRef sel = mask_indices(tc, b.selection);
  Batch o; o.schema = b.schema; o.base_rows = sel.len();
  ArrayRef filler; MemoPtr memo = std::make_shared<TakeMemo>();
  for (size_t i = 0; i < b.cols.size(); i++) {
    if (needed.count((int)i)) o.cols.push_back(col_take(b.cols[i], sel, memo));
    else { if (!filler) { dfgpu_array* f = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, DFGPU_INT8, 0, 0, o.base_rows, &f)); filler = ArrayRef::adopt(f); } o.cols.push_back(col_of(filler)); }
  }
  return o;
}

struct ProjectionExec : Plan {    // projection.rs:52-62, batch_project :295-317
  std::vector<ExprPtr> exprs; std::vector<std::string> names; PlanPtr input; mutable SchemaPtr sch; mutable std::mutex mu;
  PlanPtr fresh() const override { auto p = std::make_shared<ProjectionExec>(); p->exprs = exprs; p->names = names; p->input = input->fresh(); return p; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {input}; }
  const char* name() const override { return "ProjectionExec"; }
  SchemaPtr schema() const override {
    std::lock_guard<std::mutex> l(mu);
    if (!sch) { auto s = std::make_shared<Schema>(); auto in = input->schema();
      for (size_t i = 0; i < exprs.size(); i++) { int ci = exprs[i]->column_index(); Field f = (ci >= 0 && in && ci < (int)in->f.size()) ? in->f[(size_t)ci] : Field{}; f.name = names[i]; s->f.push_back(f); }
      sch = s; }
    return sch;
  }
  int partitions() const override { return input->partitions(); }
  struct RowSel { dfgpu_ctx* c = nullptr; ~RowSel() { if (c) dfgpu_ctx_set_row_selection(c, nullptr); } };
  bool only_columns() const { bool oc = true; for (auto& e : exprs) oc &= e->column_index() >= 0; return oc; }
  // batch_project (projection.rs:295-317).  `defer` (optional, one flag per expression) leaves computed columns unevaluated (empty Col) for
  // a consumer that can evaluate them inside its own pass (AggregateExec's fused accumulate); *deferred says whether any was left.
  Batch project(const TaskContext& tc, Batch b, std::vector<bool>* defer = nullptr) const {
    // A dense selection (>= 1/4 of the rows, e.g. TPC-H Q1's 98 %) is carried: expressions run over the full columns with the
    // selection set as the context's row selection so that dropped rows cannot raise; a sparse one is compacted first.
    // (A consumer that takes the computed columns unevaluated -- `defer`: AggregateExec -- evaluates them, if at all, under the selection as the row selection
    // (evaluate_deferred) and reads keys and arguments through the selection as a mask: nothing is evaluated here, so the selection is carried whatever its density and
    // the count -- a pass over the bitmap and a host round trip per batch -- is not taken.)
    RowSel rowsel;
    if (b.selection && !only_columns() && !defer) {
      int64_t kept = 0; tc.check(dfgpu_mask_count(tc.ctx, b.selection.a, &kept));
      if (kept * 4 >= b.base_rows) { tc.check(dfgpu_ctx_set_row_selection(tc.ctx, b.selection.a)); rowsel.c = tc.ctx; }
      else { std::set<int> need; for (auto& e : exprs) e->columns(need); b = materialize_subset(tc, b, need); if (defer) defer->assign(exprs.size(), false); defer = nullptr; }
    }
    Batch o; o.base_rows = b.base_rows; o.selection = b.selection; auto s = std::make_shared<Schema>();
    if (defer) defer->assign(exprs.size(), false);
    for (size_t i = 0; i < exprs.size(); i++) {
      int ci = exprs[i]->column_index();
      if (ci >= 0) {                 // Column = Arc clone in the reference: pass the (possibly lazy) column through
        if (ci >= (int)b.cols.size()) fail(DFGPU_INTERNAL, "PhysicalExpr Column references column at index %d but input schema only has %zu columns", ci, b.cols.size());
        o.cols.push_back(b.cols[(size_t)ci]); Field f = b.schema->f[(size_t)ci]; f.name = names[i]; s->f.push_back(f);
      } else if (defer) { (*defer)[i] = true; o.cols.push_back(Col()); s->f.push_back(Field{names[i]}); }
      else { ArrayRef a = into_array(tc, exprs[i]->eval(tc, b), b.base_rows); s->f.push_back(field_of(names[i], a.a)); o.cols.push_back(col_of(a)); }
    }
    o.schema = s; if (!defer) { std::lock_guard<std::mutex> l(mu); sch = s; }
    return o;
  }
  // evaluate one deferred expression of project() over the batch it was deferred on
  void evaluate_deferred(const TaskContext& tc, Batch& raw, Batch& projected, size_t i) const {
    RowSel rowsel;
    if (raw.selection) { tc.check(dfgpu_ctx_set_row_selection(tc.ctx, raw.selection.a)); rowsel.c = tc.ctx; }
    ArrayRef a = into_array(tc, exprs[i]->eval(tc, raw), raw.base_rows);
    projected.schema->f[i] = field_of(names[i], a.a); projected.cols[i] = col_of(a);
  }
  struct S : Stream {
    const ProjectionExec* op; std::unique_ptr<Stream> in; TaskContext tc;
    S(const ProjectionExec* o, std::unique_ptr<Stream> i, TaskContext t) : op(o), in(std::move(i)), tc(t) {}
    bool next(Batch& out) override {
      Batch b; if (!in->next(b)) return false;
      out = op->project(tc, std::move(b)); return true;
    }
  };
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override { return std::unique_ptr<Stream>(new S(this, input->run(p, tc), tc)); }
};

struct CoalesceBatchesExec : Plan {    // coalesce_batches.rs:198-260
  PlanPtr input; int64_t target;
  PlanPtr fresh() const override { auto c = std::make_shared<CoalesceBatchesExec>(); c->input = input->fresh(); c->target = target; return c; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {input}; }
  const char* name() const override { return "CoalesceBatchesExec"; }
  SchemaPtr schema() const override { return input->schema(); }
  int partitions() const override { return input->partitions(); }
  struct S : Stream {
    const CoalesceBatchesExec* op; std::unique_ptr<Stream> in; TaskContext tc; std::vector<Batch> buf; int64_t rows = 0; bool done = false;
    S(const CoalesceBatchesExec* o, std::unique_ptr<Stream> i, TaskContext t) : op(o), in(std::move(i)), tc(t) {}
    bool flush(Batch& out) { bool ok = concat_batches(tc, buf, &out); buf.clear(); rows = 0; return ok; }
    bool next(Batch& out) override {
      while (!done) {
        Batch b; if (!in->next(b)) { done = true; break; }
        if (b.selection) { if (b.base_rows >= op->target) { out = std::move(b); return true; } b = materialize(tc, b); }   // device mega-batch: keep the fused mask
        if (b.base_rows == 0) continue;
        if (b.base_rows >= op->target && buf.empty()) { out = std::move(b); return true; }
        rows += b.base_rows; buf.push_back(std::move(b));
        if (rows >= op->target) return flush(out);
      }
      if (!buf.empty()) return flush(out);
      return false;
    }
  };
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override { return std::unique_ptr<Stream>(new S(this, input->run(p, tc), tc)); }
};

struct CoalescePartitionsExec : Plan {   // coalesce_partitions.rs
  PlanPtr input;
  PlanPtr fresh() const override { auto c = std::make_shared<CoalescePartitionsExec>(); c->input = input->fresh(); return c; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {input}; }
  const char* name() const override { return "CoalescePartitionsExec"; }
  SchemaPtr schema() const override { return input->schema(); }
  int partitions() const override { return 1; }
  std::unique_ptr<Stream> execute(int, const TaskContext& tc) const override {
    std::vector<Batch> all; for (int p = 0; p < input->partitions(); p++) drain(input, p, tc, all);
    return std::unique_ptr<Stream>(new VecStream(std::move(all)));
  }
};

// ≙ BatchPartitioner::partition_iter for Partitioning::Hash (repartition/mod.rs:148-221); honours a fused selection
static void partition_batch(const TaskContext& tc, Batch& b, const std::vector<ExprPtr>& exprs, int n, std::vector<std::vector<Batch>>& outs) {
  if (b.base_rows == 0) return;
  bool safe_keys = true; for (auto& e : exprs) safe_keys &= e->safe();
  if (n <= 256 && (!b.selection || safe_keys)) {
    // one pass (dfgpu_partition_columns): the key expressions run over the full-length batch (a fused selection goes along as the mask),
    // every materialised fixed-width column is written grouped by destination in the same read; lazy / variable-width columns follow
    // through the grouped row numbers
    Batch kb = b; kb.selection = ArrayRef();
    std::vector<ArrayRef> keys; std::vector<const dfgpu_array*> kp;
    for (auto& e : exprs) { keys.push_back(into_array(tc, e->eval(tc, kb), kb.base_rows)); kp.push_back(keys.back().a); }
    std::vector<const dfgpu_array*> cp; for (auto& c : b.cols) cp.push_back(c.arr ? c.arr.a : nullptr);
    std::vector<dfgpu_array*> oc(b.cols.size(), nullptr); std::vector<int64_t> counts((size_t)n); dfgpu_array* idx = nullptr;
    tc.check(dfgpu_partition_columns(tc.ctx, kp.data(), (int32_t)kp.size(), n, cp.data(), (int32_t)cp.size(), b.selection.a, oc.data(), &idx, counts.data()));
    ArrayRef indices = ArrayRef::adopt(idx); std::vector<ArrayRef> moved; for (auto* x : oc) moved.push_back(ArrayRef::adopt(x));
    int64_t off = 0;
    for (int d = 0; d < n; d++) {
      const int64_t cnt = counts[(size_t)d];
      if (cnt) {
        auto slice = [&](const ArrayRef& a) { dfgpu_array* s = nullptr; tc.check(dfgpu_array_slice(tc.ctx, a.a, off, cnt, &s)); return ArrayRef::adopt(s); };
        ArrayRef rows;
        Batch o; o.schema = b.schema; o.base_rows = cnt; MemoPtr memo = std::make_shared<TakeMemo>();
        for (size_t c = 0; c < b.cols.size(); c++) {
          if (moved[c]) o.cols.push_back(col_of(slice(moved[c])));
          else { if (!rows) rows = slice(indices); o.cols.push_back(col_take(b.cols[c], rows, memo)); }
        }
        outs[(size_t)d].push_back(std::move(o));
      }
      off += cnt;
    }
    return;
  }
  ArrayRef sel; Batch kb = b;
  if (b.selection) { std::set<int> need; for (auto& e : exprs) e->columns(need); sel = mask_indices(tc, b.selection); kb = materialize_subset(tc, b, need); }
  if (kb.base_rows == 0) return;
  std::vector<ArrayRef> keys; std::vector<const dfgpu_array*> kp;
  for (auto& e : exprs) { keys.push_back(into_array(tc, e->eval(tc, kb), kb.base_rows)); kp.push_back(keys.back().a); }
  std::vector<int64_t> counts((size_t)n); dfgpu_array* idx = nullptr;
  tc.check(dfgpu_hash_partition(tc.ctx, kp.data(), (int32_t)kp.size(), n, &idx, counts.data()));
  ArrayRef indices = ArrayRef::adopt(idx); int64_t off = 0;
  for (int d = 0; d < n; d++) {
    if (counts[(size_t)d]) {
      dfgpu_array* s = nullptr; tc.check(dfgpu_array_slice(tc.ctx, indices.a, off, counts[(size_t)d], &s)); ArrayRef part = ArrayRef::adopt(s);
      ArrayRef rows = sel ? take(tc, sel, part) : part;
      Batch o; o.schema = b.schema; o.base_rows = counts[(size_t)d];
      MemoPtr memo = std::make_shared<TakeMemo>(); for (auto& c : b.cols) o.cols.push_back(col_take(c, rows, memo));
      outs[(size_t)d].push_back(std::move(o));
    }
    off += counts[(size_t)d];
  }
}

struct RepartitionExec : Plan {   // repartition/mod.rs:232-294; all inputs are pulled once, outputs cached per partition
  PlanPtr input; std::vector<ExprPtr> exprs; int n;
  mutable std::mutex mu; mutable bool ran = false; mutable std::vector<std::vector<Batch>> outs;
  PlanPtr fresh() const override { auto r = std::make_shared<RepartitionExec>(); r->input = input->fresh(); r->exprs = exprs; r->n = n; return r; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {input}; }
  const char* name() const override { return "RepartitionExec"; }
  SchemaPtr schema() const override { return input->schema(); }
  int partitions() const override { return n; }
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override {
    std::lock_guard<std::mutex> l(mu);
    if (!ran) {
      outs.assign((size_t)n, {});
      for (int ip = 0; ip < input->partitions(); ip++) {
        std::vector<Batch> in; drain(input, ip, tc, in);
        int rr = 0;         // every input partition has its own BatchPartitioner, next_idx starts at 0 (repartition/mod.rs:105-109, :156-160)
        for (auto& b : in) {
          if (exprs.empty()) { outs[(size_t)(rr++ % n)].push_back(materialize(tc, b)); }   // RoundRobinBatch: whole batches in rotation
          else { SpanGuard sp(tc, met.get(), 3); partition_batch(tc, b, exprs, n, outs); }      // repartition_time (repartition/mod.rs:318)
        }
      }
      ran = true;
    }
    if (p < 0 || p >= n) fail(DFGPU_INTERNAL, "RepartitionExec invalid partition %d", p);
    return std::unique_ptr<Stream>(new VecStream(outs[(size_t)p]));
  }
};

// ------------------------------------------------------------------ HashJoinExec
struct JoinTableRef { dfgpu_join_table* t = nullptr; ~JoinTableRef() { if (t) dfgpu_join_table_free(t); } };
struct BuildSide { Batch batch; std::shared_ptr<JoinTableRef> table; std::vector<int64_t> segments; bool empty = true; };

struct HashJoinExec : Plan {      // joins/hash_join.rs:283-330
  PlanPtr left, right; std::vector<ExprPtr> on_l, on_r; ExprPtr filter; std::vector<int> f_side, f_index;
  int join_type, mode; bool null_equals_null; bool swap_small_right = true;
  // set by the operator above when it fuses a selection into its own pass (another join's build or probe, a hash repartition): only then may this join answer with the probe
  // batch under a selection (dfgpu_join_probe_selection) instead of index vectors -- an aggregate or a sort above would have to compact it first and gains nothing
  mutable bool selection_consumer = false;
  mutable std::mutex mu; mutable std::shared_ptr<BuildSide> shared;     // CollectLeft: OnceAsync (joins/utils.rs:736-776)
  PlanPtr fresh() const override {
    auto j = std::make_shared<HashJoinExec>(); j->left = left->fresh(); j->right = right->fresh(); j->on_l = on_l; j->on_r = on_r; j->filter = filter; j->f_side = f_side; j->f_index = f_index;
    j->join_type = join_type; j->mode = mode; j->null_equals_null = null_equals_null; j->swap_small_right = swap_small_right; j->selection_consumer = selection_consumer; return j;
  }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {left, right}; }
  const char* name() const override { return "HashJoinExec"; }
  bool swap_allowed(const TaskContext& tc) const { int64_t v = 1; dfgpu_ctx_get_option(tc.ctx, "join_swap_small_semi", &v); return swap_small_right && v != 0; }
  bool left_only() const { return join_type == DFGPU_JOIN_LEFT_SEMI || join_type == DFGPU_JOIN_LEFT_ANTI; }
  bool right_only() const { return join_type == DFGPU_JOIN_RIGHT_SEMI || join_type == DFGPU_JOIN_RIGHT_ANTI; }
  SchemaPtr schema() const override {       // build_join_schema (joins/utils.rs:657-729)
    auto s = std::make_shared<Schema>(); auto l = left->schema(), r = right->schema();
    if (!right_only() && l) s->f.insert(s->f.end(), l->f.begin(), l->f.end());
    if (!left_only() && r) s->f.insert(s->f.end(), r->f.begin(), r->f.end());
    return s;
  }
  int partitions() const override { return right->partitions(); }
  // collect_left_input (hash_join.rs:678-768); the device table indexes the build side in ORIGINAL input order
  std::shared_ptr<BuildSide> collect_left(int partition, const TaskContext& tc, ArrayRef* fused) const {
    auto bs = std::make_shared<BuildSide>();
    std::vector<Batch> in;
    if (partition < 0) { for (int p = 0; p < left->partitions(); p++) drain(left, p, tc, in); } else drain(left, partition, tc, in);
    if (in.size() == 1 && in[0].selection) { bs->batch = in[0]; *fused = in[0].selection; bs->batch.selection = ArrayRef(); bs->segments = { bs->batch.base_rows }; bs->empty = false; }
    else {
      std::vector<Batch> m; for (auto& b : in) { Batch x = materialize(tc, b); if (x.base_rows) { bs->segments.push_back(x.base_rows); m.push_back(std::move(x)); } }
      if (m.empty()) return bs;
      concat_batches(tc, m, &bs->batch); bs->empty = false;
    }
    return bs;
  }
  void build_table(const std::shared_ptr<BuildSide>& bs, const ArrayRef& fused, const TaskContext& tc) const {
    if (bs->empty) return;
    std::vector<ArrayRef> keys; std::vector<const dfgpu_array*> kp;
    for (auto& e : on_l) { keys.push_back(into_array(tc, e->eval(tc, bs->batch), bs->batch.base_rows)); kp.push_back(keys.back().a); }
    bs->table = std::make_shared<JoinTableRef>();
    tc.check(dfgpu_join_build(tc.ctx, kp.data(), (int32_t)kp.size(), fused.a, null_equals_null ? 1 : 0, &bs->table->t));
  }
  std::shared_ptr<BuildSide> collect_build(int partition, const TaskContext& tc) const {
    SpanGuard sp(tc, met.get(), 1);            // build_time (joins/utils.rs:1370): collecting the build input + building the table
    ArrayRef fused; auto bs = collect_left(partition, tc, &fused); build_table(bs, fused, tc); return bs;
  }
  struct S : Stream {
    const HashJoinExec* op; TaskContext tc; int partition; std::unique_ptr<Stream> probe; std::shared_ptr<BuildSide> bs; Batch swapped; int state = 0;
    // reference-sized probe batches are answered in the reference's output chunks (get_matched_indices_with_limit_offset, joins/utils.rs:284-348;
    // process_probe_batch, hash_join.rs:1238-1343): `batch_size` candidate pairs per emitted batch, index alignment per chunk
    Batch chunk_pb; ArrayRef chunk_b, chunk_p; int64_t chunk_k = -1, chunk_total = 0, chunk_joined = -1; std::vector<uint32_t> chunk_last;     // chunk_last: without a JoinFilter, every chunk's last joined probe row (one read-back per probe batch)   // 0 WaitBuildSide, 1 probing, 2 final pass, 3 done, 4 swapped semi/anti result pending
    SchemaPtr out_schema;
    bool lazy_build_rows = true, selection_output = true;
    S(const HashJoinExec* o, int p, TaskContext t) : op(o), tc(t), partition(p) { int64_t v = 1; if (dfgpu_ctx_get_option(tc.ctx, "join_lazy_build_rows", &v) == DFGPU_OK) lazy_build_rows = v != 0;
      v = 1; if (dfgpu_ctx_get_option(tc.ctx, "join_selection_output", &v) == DFGPU_OK) selection_output = v != 0; }
    int64_t last_u32(const ArrayRef& a) {
      int64_t n = a.len(); if (!n) return -1;
      dfgpu_array* s1 = nullptr; tc.check(dfgpu_array_slice(tc.ctx, a.a, n - 1, 1, &s1)); ArrayRef one = ArrayRef::adopt(s1);
      uint32_t v = 0; tc.check(dfgpu_array_export_host(tc.ctx, one.a, &v, nullptr, nullptr)); return (int64_t)v;
    }
    // The probe row of the last pair of every `bsz`-pair chunk: the reference resumes the next chunk from an in-memory offset (hash_join.rs:1332-1340); here the
    // rows are gathered from the pair list at the chunk ends and read back together -- one host read per probe batch, not one per emitted chunk.
    void chunk_last_rows(const ArrayRef& pidx, int64_t bsz) {
      const int64_t m = pidx.len(); chunk_last.clear(); if (!m) return;
      std::vector<uint32_t> pos; for (int64_t e = bsz; ; e += bsz) { pos.push_back((uint32_t)((e < m ? e : m) - 1)); if (e >= m) break; }
      if (pos.size() == 1) { chunk_last.push_back((uint32_t)last_u32(pidx)); return; }
      dfgpu_array_desc d{}; d.type = DFGPU_UINT32; d.length = (int64_t)pos.size(); d.values = pos.data();
      dfgpu_array* a = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &d, &a)); ArrayRef at = ArrayRef::adopt(a);
      ArrayRef ends = take(tc, pidx, at); chunk_last.resize(pos.size());
      tc.check(dfgpu_array_export_host(tc.ctx, ends.a, chunk_last.data(), nullptr, nullptr));
    }
    ArrayRef slice_of(const ArrayRef& a, int64_t off, int64_t len) { dfgpu_array* s1 = nullptr; tc.check(dfgpu_array_slice(tc.ctx, a.a, off, len, &s1)); return ArrayRef::adopt(s1); }
    void apply_filter(Batch& pb, ArrayRef& bidx, ArrayRef& pidx) {      // apply_join_filter_to_indices (joins/utils.rs:1143-1176)
      if (!op->filter || !bidx.len()) return;
      Batch inter; inter.schema = std::make_shared<Schema>(); inter.base_rows = bidx.len();
      for (size_t i = 0; i < op->f_side.size(); i++) {
        Col src = op->f_side[i] == 0 ? bs->batch.cols.at((size_t)op->f_index[i]) : pb.cols.at((size_t)op->f_index[i]);
        Col t = col_take(src, op->f_side[i] == 0 ? bidx : pidx); inter.cols.push_back(col_of(col_get(tc, t))); inter.schema->f.push_back(Field{"x"});
      }
      ArrayRef m = into_array(tc, op->filter->eval(tc, inter), inter.base_rows);
      ArrayRef nb = filter_idx(bidx, m), np = filter_idx(pidx, m); bidx = nb; pidx = np;
    }
    // one output chunk of the current reference-sized probe batch
    void emit_chunk(Batch& out, bool need_final) {
      const int64_t bsz = tc.batch_size > 0 ? tc.batch_size : 8192, m = chunk_b.len(), off = chunk_k * bsz;
      const int64_t len = off >= m ? 0 : (m - off < bsz ? m - off : bsz);
      const bool last = chunk_k == chunk_total - 1;
      ArrayRef b = slice_of(chunk_b, off < m ? off : m, len), p = slice_of(chunk_p, off < m ? off : m, len);
      apply_filter(chunk_pb, b, p);
      if (need_final && !bs->empty && b.len()) tc.check(dfgpu_join_mark_visited(tc.ctx, bs->table->t, b.a));
      // counts as joined after the key comparison and the join filter; without a filter the chunk's last pair is known since the probe (chunk_last_rows)
      const int64_t last_joined = op->filter ? last_u32(p) : (len > 0 ? (int64_t)chunk_last[(size_t)chunk_k] : -1);
      const int64_t r0 = chunk_joined + 1, r1 = last ? chunk_pb.base_rows : (last_joined >= 0 ? last_joined + 1 : 0);
      if (!last && last_joined >= 0) chunk_joined = last_joined;
      const int jt = op->join_type;
      if (jt == DFGPU_JOIN_RIGHT || jt == DFGPU_JOIN_FULL || op->right_only()) {
        dfgpu_array *b2 = nullptr, *p2 = nullptr;
        tc.check(dfgpu_join_adjust_indices(tc.ctx, b.a, p.a, r0, r1, jt, &b2, &p2)); b = ArrayRef::adopt(b2); p = ArrayRef::adopt(p2);
      } else if (op->left_only()) { b = slice_of(b, 0, 0); p = slice_of(p, 0, 0); }
      out = build_batch(bs->empty ? nullptr : &bs->batch, chunk_pb, b, p);
      if (last) { chunk_k = -1; chunk_pb = Batch(); chunk_b = ArrayRef(); chunk_p = ArrayRef(); } else chunk_k++;
    }
    ArrayRef filter_idx(const ArrayRef& idx, const ArrayRef& m) { dfgpu_array* o = nullptr; tc.check(dfgpu_filter(tc.ctx, idx.a, m.a, &o)); return ArrayRef::adopt(o); }
    std::vector<int> key_aliases(Batch* build, Batch& probe_b) {      // build column -> the probe column that holds the same values in every output row, or -1
      auto lf = op->left->schema(); std::vector<int> alias(lf ? lf->f.size() : 0, -1);
      if (build && op->join_type == DFGPU_JOIN_INNER && !op->null_equals_null && !op->right_only() && !op->left_only())
        for (size_t k = 0; k < op->on_l.size(); k++) {
          const int bi = op->on_l[k]->column_index(), pi = op->on_r[k]->column_index();
          if (bi < 0 || pi < 0 || bi >= (int)build->cols.size() || pi >= (int)probe_b.cols.size() || bi >= (int)alias.size()) continue;
          const Col& bc = build->cols[(size_t)bi]; const Col& pc = probe_b.cols[(size_t)pi];
          const dfgpu_array* ba = bc.arr ? bc.arr.a : bc.source.a; const dfgpu_array* pa = pc.arr ? pc.arr.a : pc.source.a; if (!ba || !pa) continue;
          dfgpu_array_desc bd, pd; dfgpu_array_describe(ba, &bd); dfgpu_array_describe(pa, &pd);
          const bool exact = (bd.type >= DFGPU_INT8 && bd.type <= DFGPU_UINT64) || bd.type == DFGPU_DATE32 || bd.type == DFGPU_DECIMAL128;
          if (exact && bd.type == pd.type && bd.precision == pd.precision && bd.scale == pd.scale) alias[(size_t)bi] = pi;
        }
      return alias;
    }
    Batch build_batch(Batch* build, Batch& probe_b, const ArrayRef& bidx, const ArrayRef& pidx, const std::shared_ptr<LazyLookup>& lazy = nullptr) {     // build_batch_from_indices (joins/utils.rs:1180-1230)
      Batch o; o.schema = out_schema; o.base_rows = pidx.len(); MemoPtr memo = std::make_shared<TakeMemo>();
      auto lf = op->left->schema();
      // Inner join on plain columns of one integer / date / decimal type: in every output row the build side's key column holds the probe side's key value, so it is taken
      // from the probe column through pidx (ascending 32-bit indices; the column itself when every probe row matched once) instead of through bidx (a random 8-byte gather
      // from the build batch).  TPC-H Q18 joins 600 M lineitems to their orders and then feeds o_orderkey into the next join: that column is l_orderkey, untouched.
      std::vector<int> alias = key_aliases(build, probe_b);
      if (!op->right_only()) for (size_t i = 0; i < (lf ? lf->f.size() : 0); i++) {
        if (build && alias[i] >= 0) o.cols.push_back(col_take(probe_b.cols[(size_t)alias[i]], pidx, memo));
        else if (build && lazy) { Col c; c.source = build->cols[i].arr; c.lookup = lazy; c.memo = memo; o.cols.push_back(std::move(c)); }       // build rows not looked up yet
        else if (build) o.cols.push_back(col_take(build->cols[i], bidx, memo));
        else { dfgpu_array* nn = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, lf->f[i].type, lf->f[i].precision, lf->f[i].scale, o.base_rows, &nn)); o.cols.push_back(col_of(ArrayRef::adopt(nn))); }
      }
      if (!op->left_only()) for (auto& c : probe_b.cols) o.cols.push_back(col_take(c, pidx, memo));
      return o;
    }
    bool next(Batch& out) override {
      if (state == 0) {           // WaitBuildSide (hash_join.rs:1149-1193)
        if (op->left_only() && op->swap_allowed(tc) && (op->mode != 0 || op->right->partitions() == 1)) {
          // LeftSemi / LeftAnti keep only left rows, in ascending index of the collected left batch, so WHICH side the device table indexes
          // is not observable: with a right side much smaller than the left (TPC-H Q18: 600 M joined rows IN a few thousand order keys)
          // the table is built on the right and the collected left batch probes it once.
          ArrayRef fused; bs = op->collect_left(op->mode == 0 ? -1 : partition, tc, &fused);
          std::vector<Batch> rin; drain(op->right, partition, tc, rin);
          Batch rb; bool have_right = concat_batches(tc, rin, &rb) && rb.base_rows > 0;       // selections applied: base_rows is the row count
          out_schema = op->schema();
          if (!bs->empty && rb.base_rows * 8 <= bs->batch.base_rows) { swapped = semi_by_probing_left(fused, have_right ? &rb : nullptr); state = 4; }
          else { op->build_table(bs, fused, tc); std::vector<Batch> one; if (have_right) one.push_back(std::move(rb)); probe.reset(new VecStream(std::move(one))); state = 1; }
        }
        else if (op->mode == 0) { std::lock_guard<std::mutex> l(op->mu); if (!op->shared) op->shared = op->collect_build(-1, tc); bs = op->shared; }
        else bs = op->collect_build(partition, tc);
        if (state == 0) { probe = op->right->run(partition, tc); out_schema = op->schema(); state = 1; }
      }
      if (state == 4) { state = 3; if (swapped.base_rows == 0) return false; out = std::move(swapped); return true; }
      bool need_final = op->join_type == DFGPU_JOIN_LEFT || op->join_type == DFGPU_JOIN_FULL || op->left_only();     // need_produce_result_in_final
      // CollectLeft shares ONE build side between the probe partitions; the reference gives every stream a visited bitmap of its own
      // (hash_join.rs:1172-1190), so with several probe partitions each stream would emit unmatched build rows from a partial view.
      // That plan shape is refused rather than answered differently from either reading.
      if (need_final && op->mode == 0 && op->right->partitions() > 1 && state != 4)
        fail(DFGPU_NOT_IMPLEMENTED, "HashJoinExec mode=CollectLeft with %d probe partitions and a join type that emits build rows in a final pass; repartition both sides (Partitioned) or coalesce the probe side", op->right->partitions());
      while (state == 1) {        // FetchProbeBatch / ProcessProbeBatch (:1199-1343)
        if (chunk_k >= 0) { SpanGuard join_span(tc, op->met.get(), 2); emit_chunk(out, need_final); return true; }
        Batch pb; if (!probe->next(pb)) { state = 2; break; }
        if (pb.base_rows == 0) continue;
        SpanGuard join_span(tc, op->met.get(), 2);        // join_time (joins/utils.rs:1381): probing one batch and building its output
        // a probe batch of the reference's size (<= max(batch_size, 8192) rows) comes out in the reference's chunks; the device's own
        // whole-partition batches are answered in one piece (their consumers re-slice to batch_size)
        const bool chunked = pb.base_rows <= (tc.batch_size > 8192 ? tc.batch_size : 8192);
        if (chunked && pb.selection) { pb = materialize(tc, pb); if (pb.base_rows == 0) continue; }
        // Right / Full / RightSemi / RightAnti emit the probe rows WITHOUT a match (adjust_indices_by_join_type over the batch's row range,
        // joins/utils.rs:1234-1279): rows a fused FilterExec dropped must not come back as unmatched rows, so the selection is applied first
        if (pb.selection && (op->join_type == DFGPU_JOIN_RIGHT || op->join_type == DFGPU_JOIN_FULL || op->right_only())) { pb = materialize(tc, pb); if (pb.base_rows == 0) continue; }
        ArrayRef mask = pb.selection; pb.selection = ArrayRef();
        ArrayRef bidx, pidx;
        if (bs->empty) { dfgpu_array *a = nullptr, *b = nullptr; dfgpu_array_desc d{}; d.type = DFGPU_UINT64; d.values = &d; tc.check(dfgpu_array_import_host(tc.ctx, &d, &a)); d.type = DFGPU_UINT32; tc.check(dfgpu_array_import_host(tc.ctx, &d, &b)); bidx = ArrayRef::adopt(a); pidx = ArrayRef::adopt(b); }
        else {
          std::vector<ArrayRef> keys; std::vector<const dfgpu_array*> kp;
          for (auto& e : op->on_r) { keys.push_back(into_array(tc, e->eval(tc, pb), pb.base_rows)); kp.push_back(keys.back().a); }
          dfgpu_array *b = nullptr, *p = nullptr;
          // an Inner join whose build rows nothing here needs (no filter, no final pass, one piece): the table may leave them for later (LazyLookup)
          bool defer = !chunked && op->join_type == DFGPU_JOIN_INNER && !op->filter && !need_final && kp.size() == 1 && lazy_build_rows;
          // An Inner join over a unique build of which nothing but (aliased) key columns leaves the build side: its output is the probe batch under the selection
          // "row is selected and finds its key" -- the probe's match bits.  No index vector, no gather, no row count on the host: the consumer fuses the selection
          // (the next join's build or probe, a repartition, an aggregate) or compacts it when it must.  TPC-H Q3: customer x orders hands the orders table itself,
          // under a selection, to the build of the join with lineitem.
          if (defer && selection_output && op->selection_consumer) {
            std::vector<int> al = key_aliases(&bs->batch, pb); bool all = !al.empty(); for (int a : al) all = all && a >= 0;
            if (all) {
              dfgpu_array* s = nullptr; dfgpu_status st = dfgpu_join_probe_selection(tc.ctx, bs->table->t, kp.data(), 1, mask.a, &s);
              if (st == DFGPU_OK) {
                Batch o; o.schema = out_schema; o.base_rows = pb.base_rows; o.selection = ArrayRef::adopt(s);
                for (int a : al) o.cols.push_back(pb.cols[(size_t)a]);
                for (auto& c : pb.cols) o.cols.push_back(c);
                out = std::move(o); return true;
              }
              if (st != DFGPU_NOT_IMPLEMENTED) tc.check(st);
            }
          }
          for (auto& c : bs->batch.cols) defer = defer && (bool)c.arr;
          if (defer) {
            tc.check(dfgpu_join_probe_deferred(tc.ctx, bs->table->t, kp.data(), 1, mask.a, &b, &p)); pidx = ArrayRef::adopt(p);
            if (!b) {
              auto lz = std::make_shared<LazyLookup>(); lz->keep = bs; lz->table = bs->table->t; lz->probe_key = keys[0]; lz->rows = pidx; lz->m = pidx.len();
              // Left for later when that can only win: every probe row matched (the lookup is the big one, and a later operator may thin the rows first -- Q18), or no
              // build column but aliased key columns leaves this join (nobody will ever ask -- Q3's customer side).  Otherwise now, while keys and bitmap are in cache:
              // deferring the lookup of a 15 % match cost TPC-H Q5 0.3 ms of 7.2.
              bool wanted = false; { std::vector<int> al = key_aliases(&bs->batch, pb); for (int a : al) wanted |= a < 0; }
              if (!wanted || (dfgpu_array_is_identity(pidx.a) && pidx.len() == pb.base_rows)) { out = build_batch(&bs->batch, pb, ArrayRef(), pidx, lz); return true; }
              bidx = lz->resolve(tc, ArrayRef());
            } else bidx = ArrayRef::adopt(b);
          } else { tc.check(dfgpu_join_probe(tc.ctx, bs->table->t, kp.data(), (int32_t)kp.size(), mask.a, &b, &p)); bidx = ArrayRef::adopt(b); pidx = ArrayRef::adopt(p); }
          if (!chunked) { apply_filter(pb, bidx, pidx); if (need_final) tc.check(dfgpu_join_mark_visited(tc.ctx, bs->table->t, bidx.a)); }
        }
        if (chunked) {
          // chunks of `batch_size` pairs; one more (empty) lookup follows when the limit was hit before the scan reached the end of the batch
          // (chain_traverse!, joins/utils.rs:147-187: next_offset is None only at the last chain element of the last probe row)
          const int64_t bsz = tc.batch_size > 0 ? tc.batch_size : 8192, m = bidx.len();
          chunk_total = m == 0 ? 1 : (m + bsz - 1) / bsz;
          chunk_last_rows(pidx, bsz);
          if (m > 0 && m % bsz == 0 && (int64_t)chunk_last.back() != pb.base_rows - 1) chunk_total++;
          chunk_pb = std::move(pb); chunk_b = bidx; chunk_p = pidx; chunk_k = 0; chunk_joined = -1;
          continue;
        }
        int jt = op->join_type;
        if (jt == DFGPU_JOIN_RIGHT || jt == DFGPU_JOIN_FULL || op->right_only()) {
          dfgpu_array *b2 = nullptr, *p2 = nullptr;
          tc.check(dfgpu_join_adjust_indices(tc.ctx, bidx.a, pidx.a, 0, pb.base_rows, jt, &b2, &p2)); bidx = ArrayRef::adopt(b2); pidx = ArrayRef::adopt(p2);
        } else if (op->left_only()) continue;
        out = build_batch(bs->empty ? nullptr : &bs->batch, pb, bidx, pidx); return true;
      }
      if (state == 2) {           // ExhaustedProbeSide -> process_unmatched_build_batch (:1348-1388)
        state = 3;
        if (!need_final || bs->empty) return false;
        dfgpu_array* f = nullptr; tc.check(dfgpu_join_final_indices(tc.ctx, bs->table->t, op->join_type, &f)); ArrayRef fidx = ArrayRef::adopt(f);
        fidx = reference_final_order(fidx);
        Batch o; o.schema = out_schema; o.base_rows = fidx.len();
        MemoPtr memo = std::make_shared<TakeMemo>(); for (auto& c : bs->batch.cols) o.cols.push_back(col_take(c, fidx, memo));
        if (!op->left_only()) { auto rf = op->right->schema(); for (auto& fd : rf->f) { dfgpu_array* nn = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, fd.type, fd.precision, fd.scale, o.base_rows, &nn)); o.cols.push_back(col_of(ArrayRef::adopt(nn))); } }
        out = std::move(o); return true;
      }
      return false;
    }
    Batch semi_by_probing_left(ArrayRef fused, Batch* right_rows) {
      Batch o; o.schema = out_schema;
      const bool have_right = right_rows != nullptr; Batch none; Batch& rb = have_right ? *right_rows : none;
      Batch& lb = bs->batch;
      const bool anti = op->join_type == DFGPU_JOIN_LEFT_ANTI;
      if (anti && fused) {          // the complement below runs over every row of the left batch: apply a fused selection first
        Batch sel = lb; sel.selection = fused; lb = materialize(tc, sel); fused = ArrayRef(); bs->segments = { lb.base_rows };
      }
      ArrayRef lidx;
      if (!have_right) {
        if (!anti) return o;
        dfgpu_array *e0 = nullptr, *e1 = nullptr; dfgpu_array_desc d{}; d.type = DFGPU_UINT64; d.values = &d; tc.check(dfgpu_array_import_host(tc.ctx, &d, &e0)); ArrayRef b0 = ArrayRef::adopt(e0);
        d.type = DFGPU_UINT32; tc.check(dfgpu_array_import_host(tc.ctx, &d, &e1)); ArrayRef p0 = ArrayRef::adopt(e1);
        dfgpu_array *b2 = nullptr, *p2 = nullptr; tc.check(dfgpu_join_adjust_indices(tc.ctx, b0.a, p0.a, 0, lb.base_rows, DFGPU_JOIN_RIGHT_ANTI, &b2, &p2)); ArrayRef drop = ArrayRef::adopt(b2); lidx = ArrayRef::adopt(p2);
      } else {
        std::vector<ArrayRef> rk, lk; std::vector<const dfgpu_array*> rp, lp;
        for (auto& e : op->on_r) { rk.push_back(into_array(tc, e->eval(tc, rb), rb.base_rows)); rp.push_back(rk.back().a); }
        JoinTableRef table; tc.check(dfgpu_join_build(tc.ctx, rp.data(), (int32_t)rp.size(), nullptr, op->null_equals_null ? 1 : 0, &table.t));
        for (auto& e : op->on_l) { lk.push_back(into_array(tc, e->eval(tc, lb), lb.base_rows)); lp.push_back(lk.back().a); }
        dfgpu_array *b = nullptr, *p = nullptr;
        tc.check(dfgpu_join_probe(tc.ctx, table.t, lp.data(), (int32_t)lp.size(), fused.a, &b, &p)); ArrayRef ridx = ArrayRef::adopt(b); lidx = ArrayRef::adopt(p);
        if (op->filter && ridx.len()) {       // the filter's side 0 is still the left input
          Batch inter; inter.schema = std::make_shared<Schema>(); inter.base_rows = ridx.len();
          for (size_t i = 0; i < op->f_side.size(); i++) {
            Col src = op->f_side[i] == 0 ? lb.cols.at((size_t)op->f_index[i]) : rb.cols.at((size_t)op->f_index[i]);
            Col t = col_take(src, op->f_side[i] == 0 ? lidx : ridx); inter.cols.push_back(col_of(col_get(tc, t))); inter.schema->f.push_back(Field{"x"});
          }
          ArrayRef m = into_array(tc, op->filter->eval(tc, inter), inter.base_rows);
          ArrayRef nr = filter_idx(ridx, m), nl = filter_idx(lidx, m); ridx = nr; lidx = nl;
        }
        dfgpu_array *b2 = nullptr, *p2 = nullptr;       // get_semi_indices / get_anti_indices over the left rows (joins/utils.rs:1309-1364)
        tc.check(dfgpu_join_adjust_indices(tc.ctx, ridx.a, lidx.a, 0, lb.base_rows, anti ? DFGPU_JOIN_RIGHT_ANTI : DFGPU_JOIN_RIGHT_SEMI, &b2, &p2));
        ArrayRef drop = ArrayRef::adopt(b2); lidx = ArrayRef::adopt(p2);
      }
      if (bs->segments.size() > 1 && lidx.len()) {
        dfgpu_array* w = nullptr; tc.check(dfgpu_cast(tc.ctx, lidx.a, DFGPU_UINT64, 0, 0, &w)); lidx = reference_final_order(ArrayRef::adopt(w));
      }
      o.base_rows = lidx.len();
      MemoPtr memo = std::make_shared<TakeMemo>(); for (auto& c : lb.cols) o.cols.push_back(col_take(c, lidx, memo));
      return o;
    }
    // The reference concatenates the build batches in REVERSED order (hash_join.rs:746,764) and emits the final
    // unmatched / semi rows in ascending index of THAT batch (joins/utils.rs:1119-1141): last input batch first.
    ArrayRef reference_final_order(const ArrayRef& fidx) {
      if (bs->segments.size() <= 1 || fidx.len() == 0) return fidx;
      std::vector<int64_t> bounds{0}; for (auto s : bs->segments) bounds.push_back(bounds.back() + s);
      if (bs->segments.size() <= 64) {          // on the device: the ascending index list is cut at the batch boundaries and the pieces are put back last batch first
        auto scalar = [&](uint64_t v) { dfgpu_array_desc d{}; d.type = DFGPU_UINT64; d.length = 1; d.values = &v; dfgpu_array* a = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &d, &a)); return ArrayRef::adopt(a); };
        std::vector<ArrayRef> pieces;
        for (size_t sg = bs->segments.size(); sg-- > 0;) {
          ArrayRef lo = scalar((uint64_t)bounds[sg]), hi = scalar((uint64_t)bounds[sg + 1]); dfgpu_array *ge = nullptr, *lt = nullptr, *both = nullptr, *part = nullptr;
          tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_GTEQ, fidx.a, 0, lo.a, 1, &ge)); ArrayRef a = ArrayRef::adopt(ge);
          tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_LT, fidx.a, 0, hi.a, 1, &lt)); ArrayRef b = ArrayRef::adopt(lt);
          tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_AND, a.a, 0, b.a, 0, &both)); ArrayRef m = ArrayRef::adopt(both);
          tc.check(dfgpu_filter(tc.ctx, fidx.a, m.a, &part)); ArrayRef pc = ArrayRef::adopt(part);
          if (pc.len()) pieces.push_back(pc);
        }
        if (pieces.size() == 1) return pieces[0];
        return concat_arrays(tc, pieces);
      }
      int64_t n = fidx.len(); std::vector<uint64_t> h((size_t)n);          // very many build batches: one pass on the host
      tc.check(dfgpu_array_export_host(tc.ctx, fidx.a, h.data(), nullptr, nullptr));
      std::vector<size_t> cut{0};                                           // fidx ascends: each batch's entries are one run
      for (size_t sg = 1; sg < bounds.size(); sg++) cut.push_back((size_t)(std::lower_bound(h.begin(), h.end(), (uint64_t)bounds[sg]) - h.begin()));
      std::vector<uint64_t> o; o.reserve((size_t)n);
      for (size_t sg = bs->segments.size(); sg-- > 0;) o.insert(o.end(), h.begin() + (std::ptrdiff_t)cut[sg], h.begin() + (std::ptrdiff_t)cut[sg + 1]);
      dfgpu_array_desc d{}; d.type = DFGPU_UINT64; d.length = n; d.values = o.data();
      dfgpu_array* a = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &d, &a)); return ArrayRef::adopt(a);
    }
  };
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override { return std::unique_ptr<Stream>(new S(this, p, tc)); }
};

// ------------------------------------------------------------------ SortMergeJoinExec
// joins/sort_merge_join.rs:62-160.  Both inputs arrive sorted on the join keys and partitioned alike; output partition p joins left partition p with right
// partition p.  The streamed side (left, or right for JoinType::Right; :160-175) decides the row order: streamed rows in order, each with its matching
// buffered rows in buffered order, an unmatched streamed row of an outer join in its own place with NULLs (join_partial :968-1060).  For sorted inputs that
// is exactly: probe the streamed rows against a table of the buffered rows (pairs come out by streamed row, then buffered input order), then -- for Left /
// Right -- a stable re-order of matched + unmatched rows by streamed row.  The device does that instead of a two-cursor merge: one build, one probe and
// one stable sort over whole partitions.  Inner, Left, Right, LeftSemi, LeftAnti, RightAnti (streamed side = right, :164) and Full without a JoinFilter.
// Full = the Left join's rows in streamed order, then one batch of the buffered rows no streamed row matched, NULL-joined (the reference interleaves those
// batch by batch as its buffered cursor advances, :1001-1077; its two Full tests, :2094-2121 and :2451-2497, compare sorted rows, and so do ours).
// RightSemi (refused by the reference too, :107-111) and JoinFilters answer NotImplemented.
struct SortMergeJoinExec : Plan {
  PlanPtr left, right; std::vector<ExprPtr> on_l, on_r; int join_type = 0; bool null_equals_null = false;
  ExprPtr filter; std::vector<int> f_side, f_index;          // JoinFilter: column i of the intermediate batch = column f_index[i] of side f_side[i] (0 left, 1 right)
  PlanPtr fresh() const override { auto j = std::make_shared<SortMergeJoinExec>(); j->left = left->fresh(); j->right = right->fresh(); j->on_l = on_l; j->on_r = on_r; j->join_type = join_type; j->null_equals_null = null_equals_null; j->filter = filter; j->f_side = f_side; j->f_index = f_index; return j; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {left, right}; }
  const char* name() const override { return "SortMergeJoinExec"; }
  bool left_only() const { return join_type == DFGPU_JOIN_LEFT_SEMI || join_type == DFGPU_JOIN_LEFT_ANTI; }
  bool right_only() const { return join_type == DFGPU_JOIN_RIGHT_ANTI; }
  SchemaPtr schema() const override {
    auto s = std::make_shared<Schema>(); auto l = left->schema(), r = right->schema();
    if (!right_only() && l) s->f.insert(s->f.end(), l->f.begin(), l->f.end());
    if (!left_only() && r) s->f.insert(s->f.end(), r->f.begin(), r->f.end());
    return s;
  }
  int partitions() const override { return left->partitions(); }
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override {
    if (left->partitions() != right->partitions()) fail(DFGPU_INTERNAL, "Invalid SortMergeJoinExec, partition count mismatch %d!=%d, consider using RepartitionExec", left->partitions(), right->partitions());   // :294-300
    const bool stream_left = join_type != DFGPU_JOIN_RIGHT && join_type != DFGPU_JOIN_RIGHT_ANTI;      // :160-175
    auto collect = [&](const PlanPtr& side, Batch* out) {
      std::vector<Batch> in; drain(side, p, tc, in);
      if (!concat_batches(tc, in, out)) { out->schema = side->schema(); out->base_rows = 0; for (auto& f : out->schema->f) { dfgpu_array* nn = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, f.type, f.precision, f.scale, 0, &nn)); out->cols.push_back(col_of(ArrayRef::adopt(nn))); } }
    };
    Batch lb, rb; collect(left, &lb); collect(right, &rb);
    Batch& sb = stream_left ? lb : rb; Batch& bb = stream_left ? rb : lb;
    const auto& on_s = stream_left ? on_l : on_r; const auto& on_b = stream_left ? on_r : on_l;
    std::vector<Batch> outv;
    SpanGuard join_span(tc, met.get(), 2);
    ArrayRef bidx, sidx, unmatched_b;            // unmatched_b (Full): buffered rows no streamed row matched
    {
      std::vector<ArrayRef> bk, sk; std::vector<const dfgpu_array*> bp, sp;
      for (auto& e : on_b) { bk.push_back(into_array(tc, e->eval(tc, bb), bb.base_rows)); bp.push_back(bk.back().a); }
      for (auto& e : on_s) { sk.push_back(into_array(tc, e->eval(tc, sb), sb.base_rows)); sp.push_back(sk.back().a); }
      JoinTableRef table; tc.check(dfgpu_join_build(tc.ctx, bp.data(), (int32_t)bp.size(), nullptr, null_equals_null ? 1 : 0, &table.t));
      dfgpu_array *b = nullptr, *s = nullptr; tc.check(dfgpu_join_probe(tc.ctx, table.t, sp.data(), (int32_t)sp.size(), nullptr, &b, &s)); bidx = ArrayRef::adopt(b); sidx = ArrayRef::adopt(s);
      if (join_type == DFGPU_JOIN_FULL) {        // the table indexes the buffered side: its unvisited rows are what the Full join adds (≙ the visited bitmap of a HashJoinExec Full join)
        tc.check(dfgpu_join_mark_visited(tc.ctx, table.t, bidx.a));
        dfgpu_array* u = nullptr; tc.check(dfgpu_join_final_indices(tc.ctx, table.t, DFGPU_JOIN_LEFT_ANTI, &u)); unmatched_b = ArrayRef::adopt(u);
      }
    }
    // JoinFilter (:1156-1300), as the reference applies it: over the joined PAIRS of a chunk.  A pair that passes is an output row.  For Left / Right / Full a pair that fails is
    // ALSO an output row -- its streamed row joined with NULLs (one per failing pair, not one per streamed row without a passing pair: sort_merge_join.slt:137-147 holds the
    // reference to exactly that), and for Full a second one, NULLs joined with its buffered row.  Buffered rows count as joined by their key match alone.
    ArrayRef fail_b, fail_s; const int64_t npairs = sidx.len();
    if (filter && npairs) {
      Batch inter; inter.schema = std::make_shared<Schema>(); inter.base_rows = npairs;
      for (size_t i = 0; i < f_side.size(); i++) {
        const bool from_left = f_side[i] == 0; Batch& side = from_left ? lb : rb;
        Col t = col_take(side.cols.at((size_t)f_index[i]), from_left == stream_left ? sidx : bidx); inter.cols.push_back(col_of(col_get(tc, t))); inter.schema->f.push_back(Field{"x"});
      }
      ArrayRef m = known_mask(tc, into_array(tc, filter->eval(tc, inter), npairs));
      auto pick = [&](const ArrayRef& a, const ArrayRef& mask) { dfgpu_array* o = nullptr; tc.check(dfgpu_filter(tc.ctx, a.a, mask.a, &o)); return ArrayRef::adopt(o); };
      if (join_type != DFGPU_JOIN_INNER) { dfgpu_array* nm = nullptr; tc.check(dfgpu_not(tc.ctx, m.a, &nm)); ArrayRef notm = ArrayRef::adopt(nm); fail_b = pick(bidx, notm); fail_s = pick(sidx, notm); }
      ArrayRef pb = pick(bidx, m), ps = pick(sidx, m);
      if (join_type == DFGPU_JOIN_INNER) { bidx = pb; sidx = ps; }
      else {
        // streamed rows without a key match, from the index algebra over ALL key pairs; then passing pairs + failing pairs (buffered side NULL) + those rows
        dfgpu_array *b2 = nullptr, *s2 = nullptr; tc.check(dfgpu_join_adjust_indices(tc.ctx, bidx.a, sidx.a, 0, sb.base_rows, DFGPU_JOIN_RIGHT, &b2, &s2)); ArrayRef ab = ArrayRef::adopt(b2), as_ = ArrayRef::adopt(s2);
        dfgpu_array *ub = nullptr, *us = nullptr; tc.check(dfgpu_array_slice(tc.ctx, ab.a, npairs, ab.len() - npairs, &ub)); ArrayRef unb = ArrayRef::adopt(ub);
        tc.check(dfgpu_array_slice(tc.ctx, as_.a, npairs, as_.len() - npairs, &us)); ArrayRef uns = ArrayRef::adopt(us);
        dfgpu_array* nn = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, DFGPU_UINT64, 0, 0, fail_s.len(), &nn)); ArrayRef fnull = ArrayRef::adopt(nn);
        std::vector<ArrayRef> bv, sv;
        if (pb.len()) { bv.push_back(pb); sv.push_back(ps); }
        if (fail_s.len()) { bv.push_back(fnull); sv.push_back(fail_s); }
        if (unb.len()) { bv.push_back(unb); sv.push_back(uns); }
        if (bv.empty()) { bidx = pb; sidx = ps; } else if (bv.size() == 1) { bidx = bv[0]; sidx = sv[0]; } else { bidx = concat_arrays(tc, bv); sidx = concat_arrays(tc, sv); }
        if (sidx.len()) {                                  // back into streamed order, a row's passing pairs before its NULL-joined ones (stable)
          const dfgpu_array* kp = sidx.a; uint8_t no = 0; dfgpu_array* perm = nullptr;
          tc.check(dfgpu_sort_to_indices(tc.ctx, &kp, &no, &no, 1, -1, &perm)); ArrayRef pm = ArrayRef::adopt(perm);
          bidx = take(tc, bidx, pm); sidx = take(tc, sidx, pm);
        }
      }
    }
    if (join_type != DFGPU_JOIN_INNER && !(filter && npairs)) {
      const int as = (join_type == DFGPU_JOIN_LEFT || join_type == DFGPU_JOIN_RIGHT || join_type == DFGPU_JOIN_FULL) ? DFGPU_JOIN_RIGHT : join_type == DFGPU_JOIN_LEFT_SEMI ? DFGPU_JOIN_RIGHT_SEMI : DFGPU_JOIN_RIGHT_ANTI;
      dfgpu_array *b2 = nullptr, *s2 = nullptr; tc.check(dfgpu_join_adjust_indices(tc.ctx, bidx.a, sidx.a, 0, sb.base_rows, as, &b2, &s2)); bidx = ArrayRef::adopt(b2); sidx = ArrayRef::adopt(s2);
      if (as == DFGPU_JOIN_RIGHT && sidx.len()) {         // unmatched streamed rows back into their places: stable order by streamed row
        const dfgpu_array* kp = sidx.a; uint8_t no = 0; dfgpu_array* perm = nullptr;
        tc.check(dfgpu_sort_to_indices(tc.ctx, &kp, &no, &no, 1, -1, &perm)); ArrayRef pm = ArrayRef::adopt(perm);
        bidx = take(tc, bidx, pm); sidx = take(tc, sidx, pm);
      }
    }
    Batch o; o.schema = schema(); o.base_rows = sidx.len(); MemoPtr memo = std::make_shared<TakeMemo>();
    if (!right_only()) for (auto& c : lb.cols) o.cols.push_back(col_take(c, stream_left ? sidx : bidx, memo));
    if (!left_only()) for (auto& c : rb.cols) o.cols.push_back(col_take(c, stream_left ? bidx : sidx, memo));
    if (o.base_rows > 0) outv.push_back(std::move(o));
    if (join_type == DFGPU_JOIN_FULL && fail_b && fail_b.len() > 0) {      // Full with a filter: NULLs joined with the buffered row of every failing pair (:1262-1300)
      const int64_t m = fail_b.len(); dfgpu_array* nn = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, DFGPU_UINT32, 0, 0, m, &nn)); ArrayRef nulls = ArrayRef::adopt(nn);
      Batch u; u.schema = schema(); u.base_rows = m; MemoPtr memo2 = std::make_shared<TakeMemo>();
      for (auto& c : lb.cols) u.cols.push_back(col_take(c, nulls, memo2));
      for (auto& c : rb.cols) u.cols.push_back(col_take(c, fail_b, memo2));
      outv.push_back(std::move(u));
    }
    if (unmatched_b && unmatched_b.len() > 0) {      // Full: streamed (left) columns NULL, buffered (right) columns from the unmatched rows
      const int64_t m = unmatched_b.len(); dfgpu_array* nn = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, DFGPU_UINT32, 0, 0, m, &nn)); ArrayRef nulls = ArrayRef::adopt(nn);
      Batch u; u.schema = schema(); u.base_rows = m; MemoPtr memo2 = std::make_shared<TakeMemo>();
      for (auto& c : lb.cols) u.cols.push_back(col_take(c, nulls, memo2));
      for (auto& c : rb.cols) u.cols.push_back(col_take(c, unmatched_b, memo2));
      outv.push_back(std::move(u));
    }
    return std::unique_ptr<Stream>(new VecStream(std::move(outv)));
  }
};

// ------------------------------------------------------------------ NestedLoopJoinExec
// joins/nested_loop_join.rs:84-127.  The side named by left_is_build_side (:373-378 -- left for Right / RightSemi / RightAnti / Full, right otherwise) is
// collected once; every batch of the other side is joined with it: candidate pairs left-major (build_join_indices :405-432), the JoinFilter over the
// intermediate batch (apply_join_filter_to_indices), adjust_indices_by_join_type per outer batch (:652-708), and for Full one last batch of the left rows
// no batch matched (:505-531).  Rows and row order per outer batch are the reference's; candidate pairs are generated in runs of left rows so that a
// large cross product never sits in memory at once.
struct NestedLoopJoinExec : Plan {
  PlanPtr left, right; ExprPtr filter; std::vector<int> f_side, f_index; int join_type = 0;
  mutable std::mutex mu; mutable std::shared_ptr<Batch> inner;          // OnceFut<JoinLeftData>
  bool build_left() const { return join_type == DFGPU_JOIN_RIGHT || join_type == DFGPU_JOIN_RIGHT_SEMI || join_type == DFGPU_JOIN_RIGHT_ANTI || join_type == DFGPU_JOIN_FULL; }
  bool left_only() const { return join_type == DFGPU_JOIN_LEFT_SEMI || join_type == DFGPU_JOIN_LEFT_ANTI; }
  bool right_only() const { return join_type == DFGPU_JOIN_RIGHT_SEMI || join_type == DFGPU_JOIN_RIGHT_ANTI; }
  PlanPtr fresh() const override { auto j = std::make_shared<NestedLoopJoinExec>(); j->left = left->fresh(); j->right = right->fresh(); j->filter = filter; j->f_side = f_side; j->f_index = f_index; j->join_type = join_type; return j; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {left, right}; }
  const char* name() const override { return "NestedLoopJoinExec"; }
  SchemaPtr schema() const override {
    auto s = std::make_shared<Schema>(); auto l = left->schema(), r = right->schema();
    if (!right_only() && l) s->f.insert(s->f.end(), l->f.begin(), l->f.end());
    if (!left_only() && r) s->f.insert(s->f.end(), r->f.begin(), r->f.end());
    return s;
  }
  int partitions() const override { return build_left() ? right->partitions() : left->partitions(); }
  std::shared_ptr<Batch> collect_inner(const TaskContext& tc) const {
    std::lock_guard<std::mutex> l(mu);
    if (!inner) {
      SpanGuard sp(tc, met.get(), 1);
      const PlanPtr& side = build_left() ? left : right;
      std::vector<Batch> in; for (int p = 0; p < side->partitions(); p++) drain(side, p, tc, in);
      auto b = std::make_shared<Batch>();
      if (!concat_batches(tc, in, b.get())) {           // a side without a single batch: typed empty columns, so that outer joins still null-pad it
        b->schema = side->schema(); b->base_rows = 0;
        for (auto& f : b->schema->f) { dfgpu_array* nn = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, f.type, f.precision, f.scale, 0, &nn)); b->cols.push_back(col_of(ArrayRef::adopt(nn))); }
      }
      inner = b;
    }
    return inner;
  }
  struct S : Stream {
    const NestedLoopJoinExec* op; TaskContext tc; std::unique_ptr<Stream> outer; std::shared_ptr<Batch> in; SchemaPtr out_schema; int state = 0;
    std::vector<ArrayRef> matched_left;        // Full: per outer batch, the left rows that joined (ascending, distinct)
    S(const NestedLoopJoinExec* o, int p, TaskContext t) : op(o), tc(t) { in = op->collect_inner(tc); outer = (op->build_left() ? op->right : op->left)->run(p, tc); out_schema = op->schema(); }
    ArrayRef filter_idx(const ArrayRef& idx, const ArrayRef& m) { dfgpu_array* o = nullptr; tc.check(dfgpu_filter(tc.ctx, idx.a, m.a, &o)); return ArrayRef::adopt(o); }
    ArrayRef cast_to(const ArrayRef& a, int32_t t) { dfgpu_array* o = nullptr; tc.check(dfgpu_cast(tc.ctx, a.a, t, 0, 0, &o)); return ArrayRef::adopt(o); }
    ArrayRef empty_idx(int32_t t) { dfgpu_array* o = nullptr; tc.check(dfgpu_array_new_zeros(tc.ctx, t, 0, 0, 0, &o)); return ArrayRef::adopt(o); }
    ArrayRef cat(std::vector<ArrayRef>& v, int32_t t) { if (v.empty()) return empty_idx(t); if (v.size() == 1) return v[0]; return concat_arrays(tc, v); }
    Batch build_batch(Batch& lb, Batch& rb, const ArrayRef& li, const ArrayRef& ri, int64_t rows) {
      Batch o; o.schema = out_schema; o.base_rows = rows; MemoPtr memo = std::make_shared<TakeMemo>();
      if (!op->right_only()) for (auto& c : lb.cols) o.cols.push_back(col_take(c, li, memo));
      if (!op->left_only()) for (auto& c : rb.cols) o.cols.push_back(col_take(c, ri, memo));
      return o;
    }
    // all filtered pairs of one (left batch, right batch), left-major
    void pairs(Batch& lb, Batch& rb, ArrayRef* li, ArrayRef* ri) {
      const bool bl = op->build_left(); const int64_t nl = lb.base_rows, nr = rb.base_rows;
      std::vector<ArrayRef> ls, rs;
      const int64_t run = nr > 0 ? std::max<int64_t>(1, ((int64_t)1 << 25) / nr) : nl;
      for (int64_t first = 0; first < nl && nr > 0; first += run) {
        const int64_t cnt = std::min(run, nl - first);
        dfgpu_array *a = nullptr, *b = nullptr; tc.check(dfgpu_cross_join_indices(tc.ctx, first, cnt, nr, bl ? 1 : 0, &a, &b));
        ArrayRef l = ArrayRef::adopt(a), r = ArrayRef::adopt(b);
        if (op->filter) {
          Batch inter; inter.schema = std::make_shared<Schema>(); inter.base_rows = l.len();
          for (size_t i = 0; i < op->f_side.size(); i++) {
            Col src = op->f_side[i] == 0 ? lb.cols.at((size_t)op->f_index[i]) : rb.cols.at((size_t)op->f_index[i]);
            Col t = col_take(src, op->f_side[i] == 0 ? l : r); inter.cols.push_back(col_of(col_get(tc, t))); inter.schema->f.push_back(Field{"x"});
          }
          ArrayRef m = into_array(tc, op->filter->eval(tc, inter), inter.base_rows);
          l = filter_idx(l, m); r = filter_idx(r, m);
        }
        if (l.len()) { ls.push_back(l); rs.push_back(r); }
      }
      *li = cat(ls, bl ? DFGPU_UINT64 : DFGPU_UINT32); *ri = cat(rs, bl ? DFGPU_UINT32 : DFGPU_UINT64);
    }
    bool next(Batch& out) override {
      const int jt = op->join_type; const bool bl = op->build_left();
      while (state == 0) {
        Batch ob; if (!outer->next(ob)) { state = 1; break; }
        ob = materialize(tc, ob);
        SpanGuard join_span(tc, op->met.get(), 2);
        Batch& lb = bl ? *in : ob; Batch& rb = bl ? ob : *in;
        ArrayRef li, ri; pairs(lb, rb, &li, &ri);
        if (jt == DFGPU_JOIN_FULL && li.len()) {          // visited_left_side.set_bit (:620-625)
          ArrayRef l32 = cast_to(li, DFGPU_UINT32), r64 = cast_to(ri, DFGPU_UINT64); dfgpu_array *b2 = nullptr, *p2 = nullptr;
          tc.check(dfgpu_join_adjust_indices(tc.ctx, r64.a, l32.a, 0, lb.base_rows, DFGPU_JOIN_RIGHT_SEMI, &b2, &p2)); ArrayRef drop = ArrayRef::adopt(b2); matched_left.push_back(ArrayRef::adopt(p2));
        }
        if (jt != DFGPU_JOIN_INNER) {                      // adjust_indices_by_join_type (:652-708): the streamed side plays the probe side of the hash join's index algebra
          dfgpu_array *b2 = nullptr, *p2 = nullptr;
          if (bl) { tc.check(dfgpu_join_adjust_indices(tc.ctx, li.a, ri.a, 0, rb.base_rows, jt == DFGPU_JOIN_FULL ? DFGPU_JOIN_RIGHT : jt, &b2, &p2)); li = ArrayRef::adopt(b2); ri = ArrayRef::adopt(p2); }
          else {
            int as = jt == DFGPU_JOIN_LEFT ? DFGPU_JOIN_RIGHT : jt == DFGPU_JOIN_LEFT_SEMI ? DFGPU_JOIN_RIGHT_SEMI : DFGPU_JOIN_RIGHT_ANTI;
            tc.check(dfgpu_join_adjust_indices(tc.ctx, ri.a, li.a, 0, lb.base_rows, as, &b2, &p2)); ri = ArrayRef::adopt(b2); li = ArrayRef::adopt(p2);
          }
        }
        const int64_t rows = (op->right_only() ? ri : li).len();
        out = build_batch(lb, rb, li, ri, rows); return true;           // one output batch per outer batch, empty ones included (the reference's stream does the same)
      }
      if (state == 1) {
        state = 2;
        if (jt == DFGPU_JOIN_FULL) {                      // get_final_indices_from_bit_map: left rows no right batch matched, right side NULL
          SpanGuard join_span(tc, op->met.get(), 2);
          ArrayRef all = cat(matched_left, DFGPU_UINT32), all64 = cast_to(all, DFGPU_UINT64); dfgpu_array *b2 = nullptr, *p2 = nullptr;
          tc.check(dfgpu_join_adjust_indices(tc.ctx, all64.a, all.a, 0, in->base_rows, DFGPU_JOIN_RIGHT_ANTI, &b2, &p2)); ArrayRef drop = ArrayRef::adopt(b2); ArrayRef un = ArrayRef::adopt(p2);
          Batch o; o.schema = out_schema; o.base_rows = un.len(); MemoPtr memo = std::make_shared<TakeMemo>();
          for (auto& c : in->cols) o.cols.push_back(col_take(c, un, memo));
          auto rf = op->right->schema();
          for (size_t i = 0; i < (rf ? rf->f.size() : 0); i++) { dfgpu_array* nn = nullptr; tc.check(dfgpu_array_new_null(tc.ctx, rf->f[i].type, rf->f[i].precision, rf->f[i].scale, o.base_rows, &nn)); o.cols.push_back(col_of(ArrayRef::adopt(nn))); }
          out = std::move(o); return true;
        }
      }
      return false;
    }
  };
  std::unique_ptr<Stream> execute(int p, const TaskContext& tc) const override {
    if (p < 0 || p >= partitions()) fail(DFGPU_INTERNAL, "NestedLoopJoinExec invalid partition %d", p);
    if (join_type == DFGPU_JOIN_FULL && right->partitions() > 1) fail(DFGPU_EXECUTION, "Plan error: NestedLoopJoinExec Full requires single partitions on both sides");   // distribution_from_join_type (:312-333)
    return std::unique_ptr<Stream>(new S(this, p, tc));
  }
};

// ------------------------------------------------------------------ AggregateExec
struct AggExpr { int kind; ExprPtr arg, filter; std::string name; int32_t type, precision, scale; };    // ≙ AggregateExpr for Sum/Avg/Count/Min/Max
struct GroupsRef { dfgpu_groups* g = nullptr; ~GroupsRef() { if (g) dfgpu_groups_free(g); } };
struct AccRef { dfgpu_acc* a = nullptr; ~AccRef() { if (a) dfgpu_acc_free(a); } };
static const char* agg_fun_name(int k) { switch (k) { case DFGPU_AGG_SUM: return "sum"; case DFGPU_AGG_AVG: return "avg"; case DFGPU_AGG_COUNT: return "count"; case DFGPU_AGG_MIN: return "min"; case DFGPU_AGG_COUNT_DISTINCT: return "count distinct"; default: return "max"; } }

// ---- aggregates without a GroupsAccumulator of their own in the kernel library, composed from its other entry points
// MIN / MAX over Utf8 (physical-expr/src/aggregate/min_max.rs: the row-at-a-time Accumulator behind GroupsAccumulatorAdapter): the state is one (group, value) row per
// group seen with a non-NULL value; a batch is folded in by concatenating its (group id, value) rows with the state, ordering by (group id, value) and keeping
// every group's first row.  The state of Partial is the value itself (min_max.rs state() = [evaluate()]), so every mode works.
struct StringMinMax {
  bool is_max = false; ArrayRef gids, vals;
  static ArrayRef true1(const TaskContext& tc) { uint64_t one = 1; dfgpu_array_desc d{}; d.type = DFGPU_BOOL; d.length = 1; d.values = &one; dfgpu_array* a = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &d, &a)); return ArrayRef::adopt(a); }
  static ArrayRef slice(const TaskContext& tc, const ArrayRef& a, int64_t off, int64_t len) { dfgpu_array* o = nullptr; tc.check(dfgpu_array_slice(tc.ctx, a.a, off, len, &o)); return ArrayRef::adopt(o); }
  void reduce(const TaskContext& tc, ArrayRef g, ArrayRef v) {
    const int64_t n = g.len();
    if (n == 0) { gids = g; vals = v; return; }
    const dfgpu_array* kp[2] = { g.a, v.a }; uint8_t desc[2] = { 0, (uint8_t)(is_max ? 1 : 0) }, nf[2] = { 0, 0 }; dfgpu_array* ix = nullptr;
    tc.check(dfgpu_sort_to_indices(tc.ctx, kp, desc, nf, 2, -1, &ix)); ArrayRef order = ArrayRef::adopt(ix);
    ArrayRef gs = take(tc, g, order), vs = take(tc, v, order), head = true1(tc);
    if (n > 1) {
      ArrayRef a = slice(tc, gs, 1, n - 1), b = slice(tc, gs, 0, n - 1); dfgpu_array* ne = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_NEQ, a.a, 0, b.a, 0, &ne)); ArrayRef neq = ArrayRef::adopt(ne);
      std::vector<ArrayRef> parts{head, neq}; head = concat_arrays(tc, parts);
    }
    ArrayRef hi = mask_indices(tc, head); gids = take(tc, gs, hi); vals = take(tc, vs, hi);
  }
  void update(const TaskContext& tc, const ArrayRef& g_in, ArrayRef v_in, const ArrayRef& filter) {
    dfgpu_array_desc d; dfgpu_array_describe(v_in.a, &d);
    if (d.type == DFGPU_DICTIONARY) { dfgpu_array* c = nullptr; tc.check(dfgpu_cast(tc.ctx, v_in.a, DFGPU_UTF8, 0, 0, &c)); v_in = ArrayRef::adopt(c); }
    dfgpu_array* nn = nullptr; tc.check(dfgpu_is_null(tc.ctx, v_in.a, 1, &nn)); ArrayRef mask = ArrayRef::adopt(nn);
    if (filter) { ArrayRef kf = known_mask(tc, filter); dfgpu_array* o = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_AND, mask.a, 0, kf.a, 0, &o)); mask = ArrayRef::adopt(o); }
    dfgpu_array *fg = nullptr, *fv = nullptr; tc.check(dfgpu_filter(tc.ctx, g_in.a, mask.a, &fg)); ArrayRef g = ArrayRef::adopt(fg); tc.check(dfgpu_filter(tc.ctx, v_in.a, mask.a, &fv)); ArrayRef v = ArrayRef::adopt(fv);
    if (gids && gids.len()) { std::vector<ArrayRef> gp{gids, g}, vp{vals, v}; g = concat_arrays(tc, gp); v = concat_arrays(tc, vp); }
    reduce(tc, g, v);
  }
  ArrayRef emit(const TaskContext& tc, int64_t total) {          // one row per group 0 .. total-1, NULL where no value was seen: a NULL candidate per group orders last
    dfgpu_array *io = nullptr, *nl = nullptr; tc.check(dfgpu_array_iota(tc.ctx, total, &io)); ArrayRef g = ArrayRef::adopt(io); tc.check(dfgpu_array_new_null(tc.ctx, DFGPU_UTF8, 0, 0, total, &nl)); ArrayRef v = ArrayRef::adopt(nl);
    if (gids && gids.len()) { std::vector<ArrayRef> gp{gids, g}, vp{vals, v}; g = concat_arrays(tc, gp); v = concat_arrays(tc, vp); }
    StringMinMax t; t.is_max = is_max; t.reduce(tc, g, v);
    if (t.vals.len() != total) fail(DFGPU_INTERNAL, "string MIN/MAX: %lld rows for %lld groups", (long long)t.vals.len(), (long long)total);
    return t.vals;
  }
};
// COUNT(DISTINCT x) (physical-expr/src/aggregate/count_distinct/: a set of values per group): the (group id, value) pairs are interned in a GroupValues of their own;
// every NEW pair adds one to its group's count.  Partial emits the reference's state -- one List of distinct values per group -- in the Utf8 layout (state()), Final /
// FinalPartitioned merge such lists (merge()); fixed-width arguments as packed values, Utf8 arguments as (length, bytes) strings (count_distinct/bytes.rs:47-75).
struct CountDistinct {
  GroupsRef pairs; AccRef cnt;
  void init(const TaskContext& tc) { tc.check(dfgpu_groups_new(tc.ctx, 2, &pairs.g)); tc.check(dfgpu_acc_new(tc.ctx, DFGPU_AGG_COUNT, DFGPU_INT64, 0, 0, &cnt.a)); }
  void update(const TaskContext& tc, const ArrayRef& g_in, const ArrayRef& v_in, const ArrayRef& filter, int64_t total) {
    dfgpu_array* nn = nullptr; tc.check(dfgpu_is_null(tc.ctx, v_in.a, 1, &nn)); ArrayRef mask = ArrayRef::adopt(nn);
    if (filter) { ArrayRef kf = known_mask(tc, filter); dfgpu_array* o = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_AND, mask.a, 0, kf.a, 0, &o)); mask = ArrayRef::adopt(o); }
    const int64_t before = dfgpu_groups_len(pairs.g);
    const dfgpu_array* kp[2] = { g_in.a, v_in.a }; dfgpu_array* ids = nullptr; tc.check(dfgpu_groups_intern(tc.ctx, pairs.g, kp, 2, mask.a, &ids)); ArrayRef drop = ArrayRef::adopt(ids);
    const int64_t fresh = dfgpu_groups_len(pairs.g) - before;
    if (fresh > 0) {
      dfgpu_array* keys[2] = { nullptr, nullptr }; tc.check(dfgpu_groups_emit(tc.ctx, pairs.g, keys)); ArrayRef k0 = ArrayRef::adopt(keys[0]), k1 = ArrayRef::adopt(keys[1]);
      ArrayRef gk = StringMinMax::slice(tc, k0, before, fresh);
      tc.check(dfgpu_acc_update_batch(tc.ctx, cnt.a, nullptr, gk.a, nullptr, total));
    }
  }
  ArrayRef emit(const TaskContext& tc, int64_t total) {
    dfgpu_array_desc ed{}; ed.type = DFGPU_UINT32; ed.values = &ed; dfgpu_array* e = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &ed, &e)); ArrayRef empty_ids = ArrayRef::adopt(e);
    tc.check(dfgpu_acc_update_batch(tc.ctx, cnt.a, nullptr, empty_ids.a, nullptr, total));
    dfgpu_array* v = nullptr; tc.check(dfgpu_acc_evaluate(tc.ctx, cnt.a, &v)); return ArrayRef::adopt(v);
  }
  // state() (count_distinct/native.rs:97-110: one List of the group's distinct values): the interned (group, value) pairs ordered by group -- stable, so a group's values
  // stay in first-seen order -- and cut by the groups' counts.  The list column travels in the Utf8 layout (dfgpu_list_from_counts).
  ArrayRef state(const TaskContext& tc, int64_t total, int32_t value_type) {
    ArrayRef counts = emit(tc, total);
    const int64_t np = dfgpu_groups_len(pairs.g);
    ArrayRef vals;
    if (np > 0) {
      dfgpu_array* keys[2] = { nullptr, nullptr }; tc.check(dfgpu_groups_emit(tc.ctx, pairs.g, keys)); ArrayRef k0 = ArrayRef::adopt(keys[0]), k1 = ArrayRef::adopt(keys[1]);
      const dfgpu_array* kp = k0.a; uint8_t no = 0; dfgpu_array* perm = nullptr; tc.check(dfgpu_sort_to_indices(tc.ctx, &kp, &no, &no, 1, -1, &perm)); ArrayRef pm = ArrayRef::adopt(perm);
      vals = take(tc, k1, pm);
    } else { dfgpu_array* z = nullptr; tc.check(dfgpu_array_new_zeros(tc.ctx, value_type, 0, 0, 0, &z)); vals = ArrayRef::adopt(z); }
    dfgpu_array* l = nullptr; tc.check(dfgpu_list_from_counts(tc.ctx, counts.a, vals.a, &l)); return ArrayRef::adopt(l);
  }
  // merge_batch (native.rs:131-150: every value of every incoming list is inserted into its group's set): the lists flattened, each value with the group id of its row
  void merge(const TaskContext& tc, const ArrayRef& g_in, const ArrayRef& lists, const ArrayRef& row_mask, int64_t total, int32_t value_type, int32_t precision, int32_t scale) {
    dfgpu_array *v = nullptr, *r = nullptr; tc.check(dfgpu_list_flatten(tc.ctx, lists.a, value_type, precision, scale, &v, &r)); ArrayRef vals = ArrayRef::adopt(v), row_of = ArrayRef::adopt(r);
    if (vals.len() == 0) return;                        // nothing to insert; emit() grows the counts to `total` groups
    ArrayRef ge = take(tc, g_in, row_of), fe; if (row_mask) fe = take(tc, row_mask, row_of);
    update(tc, ge, vals, fe, total);
  }
};

// what a spill keeps in host memory: Arrow arrays (dfgpu_array_export_arrow), one piece per key range of a sorted run (AggregateExec's state, SortExec's input)
struct HostColumn { ArrowArray a{}; ArrowSchema s{}; bool live = false;
  HostColumn() = default; HostColumn(const HostColumn&) = delete; HostColumn& operator=(const HostColumn&) = delete;
  HostColumn(HostColumn&& o) noexcept : a(o.a), s(o.s), live(o.live) { o.live = false; }
  ~HostColumn() { if (live) { if (a.release) a.release(&a); if (s.release) s.release(&s); } } };
struct SpillPiece { std::vector<HostColumn> cols; int64_t rows = 0; };
struct SpillRun { std::vector<SpillPiece> pieces; };

struct AggregateExec : Plan {     // aggregates/mod.rs:242-269; GroupedHashAggregateStream row_hash.rs:423-662
  int mode; std::vector<ExprPtr> gexprs; std::vector<std::string> gnames; std::vector<AggExpr> aggs; PlanPtr input; mutable SchemaPtr sch; mutable std::mutex mu;
  // PhysicalGroupBy grouping sets (aggregates/mod.rs:103-160): sets[s][i] != 0 = key i is replaced by null_exprs[i] in set s; empty = the single set of all keys
  std::vector<ExprPtr> null_exprs; std::vector<std::vector<uint8_t>> sets;
  // InputOrderMode (physical-plan/src/ordering.rs:33-44): 0 Linear, 1 PartiallySorted(order_indices: the group keys, in order, that the input is sorted on), 2 Sorted -> GroupOrdering (aggregates/order/mod.rs)
  int order_mode = 0; std::vector<int> order_indices;
  // set while the plan is built (dfgpu_plan_sort): the consumer is a SortExec over all group columns (through operators that keep rows as they are), so the order in
  // which the groups leave cannot show in the result -- the pre-aggregation skips restoring first-seen order (DFGPU_PREAGG_ANY_ORDER)
  mutable bool any_group_order = false;
  PlanPtr fresh() const override { auto a = std::make_shared<AggregateExec>(); a->any_group_order = any_group_order; a->mode = mode; a->gexprs = gexprs; a->gnames = gnames; a->aggs = aggs; a->input = input->fresh(); a->null_exprs = null_exprs; a->sets = sets; a->order_mode = order_mode; a->order_indices = order_indices; return a; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {input}; }
  const char* name() const override { return "AggregateExec"; }
  bool merging() const { return mode == 1 || mode == 2; }
  bool is_string_minmax(size_t i) const { return (aggs[i].kind == DFGPU_AGG_MIN || aggs[i].kind == DFGPU_AGG_MAX) && aggs[i].type == DFGPU_UTF8; }
  bool special(size_t i) const { return aggs[i].kind == DFGPU_AGG_COUNT_DISTINCT || is_string_minmax(i); }
  bool any_special() const { for (size_t i = 0; i < aggs.size(); i++) if (special(i)) return true; return false; }
  // The accumulator arguments as ONE expression DAG over plain columns (common subexpressions shared; references to deferred projection
  // columns expand into the projection's expression over ITS input) handed to dfgpu_acc_update_batch_fused.  false = shape not taken,
  // nothing was accumulated: the caller evaluates the arguments node by node.
  struct DagBuilder {
    const TaskContext& tc; const ProjectionExec* pj; Batch& raw; Batch& proj; const std::vector<bool>& deferred;
    std::vector<dfgpu_expr_node> nodes; std::vector<ArrayRef> keep; std::vector<const dfgpu_array*> cols; std::map<std::tuple<int, int, int>, int> seen;
    int leaf(int kind, ArrayRef a) {
      int ci = -1; for (size_t i = 0; i < cols.size(); i++) if (cols[i] == a.a) ci = (int)i;
      if (ci < 0) { ci = (int)cols.size(); cols.push_back(a.a); keep.push_back(a); }
      return node(kind, ci, 0);
    }
    int node(int op, int l, int r) {
      auto key = std::make_tuple(op, l, r); auto it = seen.find(key); if (it != seen.end()) return it->second;
      nodes.push_back(dfgpu_expr_node{op, l, r}); seen[key] = (int)nodes.size() - 1; return (int)nodes.size() - 1;
    }
    int build(const Expr* e, bool over_raw) {
      if (auto* c = dynamic_cast<const ColumnExpr*>(e)) {
        Batch& src = over_raw ? raw : proj;
        if (c->index < 0 || c->index >= (int)src.cols.size()) return -1;
        if (!over_raw && c->index < (int)deferred.size() && deferred[(size_t)c->index]) return build(pj->exprs[(size_t)c->index].get(), true);
        return leaf(DFGPU_NODE_COLUMN, src.column(tc, c->index));
      }
      if (auto* l = dynamic_cast<const LiteralExpr*>(e)) return leaf(DFGPU_NODE_SCALAR, l->scalar);
      if (auto* x = dynamic_cast<const BinaryExpr*>(e)) {
        if (x->op != DFGPU_OP_ADD && x->op != DFGPU_OP_SUB && x->op != DFGPU_OP_MUL) return -1;
        int a = build(x->l.get(), over_raw); if (a < 0) return -1;
        int b2 = build(x->r.get(), over_raw); if (b2 < 0) return -1;
        return node(x->op, a, b2);
      }
      return -1;
    }
  };
  bool try_fused(const TaskContext& tc, const ProjectionExec* pj, Batch& raw, Batch& b, const std::vector<bool>& deferred, std::vector<AccRef>& accs,
                 const ArrayRef& gids, const ArrayRef& filt, int64_t total) const {
    bool computed = false;          // worth it only when some argument is an expression, not a stored column
    for (auto& a : aggs) { if (a.filter || (a.kind != DFGPU_AGG_SUM && a.kind != DFGPU_AGG_AVG && a.kind != DFGPU_AGG_COUNT)) return false;
      if (a.arg) { int ci = a.arg->column_index(); computed |= ci < 0 || (ci < (int)deferred.size() && deferred[(size_t)ci]); } }
    if (!computed) return false;
    DagBuilder d{tc, pj, raw, b, deferred};
    std::vector<int32_t> acc_nodes; std::vector<dfgpu_acc*> ap;
    for (size_t i = 0; i < aggs.size(); i++) {
      int nd = -1;
      if (aggs[i].arg && aggs[i].kind != DFGPU_AGG_COUNT) { nd = d.build(aggs[i].arg.get(), false); if (nd < 0) return false; }
      else if (aggs[i].arg) return false;      // COUNT(expr) counts non-NULL values: leave it to the ordinary path
      acc_nodes.push_back(nd); ap.push_back(accs[i].a);
    }
    dfgpu_status st = dfgpu_acc_update_batch_fused(tc.ctx, ap.data(), acc_nodes.data(), (int32_t)ap.size(), d.nodes.data(), (int32_t)d.nodes.size(), d.cols.data(), (int32_t)d.cols.size(), gids.a, filt.a, total);
    if (st == DFGPU_NOT_IMPLEMENTED) return false;
    tc.check(st); return true;
  }
  // dfgpu_agg_preaggregate over one batch, then intern + merge_batch of its partial rows.  false = shape not taken, nothing accumulated.
  template <typename Ensure>
  bool preaggregate(const TaskContext& tc, const ProjectionExec* pj, Batch& raw, Batch& b, const std::vector<bool>& deferred, Ensure&& ensure, const std::vector<const dfgpu_array*>& keyv, const ArrayRef& mask, GroupsRef& groups, std::vector<AccRef>& accs, ArrayRef* pending) const {
    const int32_t nk = (int32_t)keyv.size();
    std::vector<ArrayRef> vals(aggs.size()); std::vector<const dfgpu_array*> vp; std::vector<int32_t> kinds;
    for (auto& a : aggs) if (a.filter) return false;
    // the key column decides (type, clustering, number of groups): ask before any computed argument is evaluated for it
    dfgpu_status v0 = dfgpu_agg_preaggregate(tc.ctx, keyv.data(), nk, nullptr, nullptr, 0, mask.a, nullptr, nullptr);
    if (v0 == DFGPU_NOT_IMPLEMENTED) return false;
    tc.check(v0);
    // An argument that is CAST(<stored integer column> AS DOUBLE) -- written as such, or a column of the projection below that was left unevaluated -- goes down as the
    // integer column with the cast named beside it (value_casts): the partition converts it while it moves it, the cast pass does not run.
    auto cast_of_column = [&](const Expr* e, Batch** src) -> int {          // -> column index in *src, or -1
      const auto* u = dynamic_cast<const UnaryExpr*>(e);
      if (!u || u->kind != 4 || u->a0 != DFGPU_FLOAT64) return -1;
      const int ci = u->e->column_index(); if (ci < 0 || ci >= (int)(*src)->cols.size()) return -1;
      if (*src == &b && ci < (int)deferred.size() && deferred[(size_t)ci]) return -1;
      return ci;
    };
    std::vector<int32_t> casts(aggs.size(), 0); bool any_cast = false;
    for (size_t i = 0; i < aggs.size(); i++) {
      if (aggs[i].arg) {
        Batch* src = &b; int ci = cast_of_column(aggs[i].arg.get(), &src);
        if (ci < 0 && pj) {          // Column(j) of the projection whose expression j is the cast, still unevaluated
          const int j = aggs[i].arg->column_index();
          if (j >= 0 && j < (int)deferred.size() && deferred[(size_t)j] && j < (int)pj->exprs.size()) { src = &raw; ci = cast_of_column(pj->exprs[(size_t)j].get(), &src); }
        }
        if (ci >= 0) {
          ArrayRef col = src->column(tc, ci); dfgpu_array_desc d; dfgpu_array_describe(col.a, &d);
          const bool is_int = d.type == DFGPU_INT32 || d.type == DFGPU_INT64;
          if (is_int && (aggs[i].kind == DFGPU_AGG_SUM || aggs[i].kind == DFGPU_AGG_AVG || aggs[i].kind == DFGPU_AGG_MIN || aggs[i].kind == DFGPU_AGG_MAX || aggs[i].kind == DFGPU_AGG_COUNT)) { vals[i] = col; casts[i] = DFGPU_FLOAT64; any_cast = true; }
        }
        if (!casts[i]) {
          std::set<int> need; aggs[i].arg->columns(need); for (int ci2 : need) ensure(ci2);
          vals[i] = into_array(tc, aggs[i].arg->eval(tc, b), b.base_rows);
        }
      }
      vp.push_back(vals[i].a); kinds.push_back(aggs[i].kind);
    }
    std::vector<dfgpu_array*> st(aggs.size() * 2 + 2, nullptr); dfgpu_array* pk[4] = { nullptr, nullptr, nullptr, nullptr };
    dfgpu_status rc = dfgpu_agg_preaggregate_flags(tc.ctx, keyv.data(), nk, kinds.data(), vp.data(), any_cast ? casts.data() : nullptr, (int32_t)aggs.size(), mask.a, any_group_order ? DFGPU_PREAGG_ANY_ORDER : 0, pk, st.data());
    if (rc == DFGPU_NOT_IMPLEMENTED && any_cast) {          // declined with the casts (a shape limit): the arguments as arrays, as before
      vp.clear();
      for (size_t i = 0; i < aggs.size(); i++) { if (casts[i]) { std::set<int> need; aggs[i].arg->columns(need); for (int ci2 : need) ensure(ci2); vals[i] = into_array(tc, aggs[i].arg->eval(tc, b), b.base_rows); } vp.push_back(vals[i].a); }
      rc = dfgpu_agg_preaggregate_flags(tc.ctx, keyv.data(), nk, kinds.data(), vp.data(), nullptr, (int32_t)aggs.size(), mask.a, any_group_order ? DFGPU_PREAGG_ANY_ORDER : 0, pk, st.data());
    }
    if (rc == DFGPU_NOT_IMPLEMENTED) return false;
    tc.check(rc);
    std::vector<ArrayRef> pkeyv; for (int32_t c = 0; c < nk; c++) pkeyv.push_back(ArrayRef::adopt(pk[c]));
    const ArrayRef& pkeys = pkeyv[0]; std::vector<ArrayRef> states; for (auto* x : st) states.push_back(ArrayRef::adopt(x));
    // A FIRST batch whose partial rows hold every key once (in first-seen order) needs no hash table to number its groups: ids are 0, 1, ..; the keys wait
    // in `pending` and are interned only if another batch follows (merge_partial with pending == nullptr), else they are emitted as they are.
    int64_t distinct = 0, fs = 1; dfgpu_ctx_get_option(tc.ctx, "agg_preaggregate_distinct", &distinct); dfgpu_ctx_get_option(tc.ctx, "first_seen_group_order", &fs); distinct = distinct && (fs || any_group_order);
    if (nk == 1 && pending && distinct && dfgpu_groups_len(groups.g) == 0 && !*pending) { *pending = pkeys; merge_partial(tc, {}, states, pkeys.len(), groups, accs); return true; }
    merge_partial(tc, pkeyv, states, 0, groups, accs);
    return true;
  }
  // intern the partial rows' keys (or take ids 0 .. n-1 when `keys` is empty) and merge their states
  void merge_partial(const TaskContext& tc, const std::vector<ArrayRef>& keys, const std::vector<ArrayRef>& states, int64_t n_ids, GroupsRef& groups, std::vector<AccRef>& accs) const {
    dfgpu_array* ids = nullptr; int64_t total;
    if (!keys.empty()) { std::vector<const dfgpu_array*> kp; for (auto& k : keys) kp.push_back(k.a); tc.check(dfgpu_groups_intern(tc.ctx, groups.g, kp.data(), (int32_t)kp.size(), nullptr, &ids)); total = dfgpu_groups_len(groups.g); }
    else { tc.check(dfgpu_array_iota(tc.ctx, n_ids, &ids)); total = n_ids; }
    ArrayRef gids = ArrayRef::adopt(ids);
    for (size_t i = 0; i < aggs.size(); i++) {
      const dfgpu_array* sp[2] = { states[2 * i].a, states[2 * i + 1].a };
      tc.check(dfgpu_acc_merge_batch(tc.ctx, accs[i].a, sp, aggs[i].kind == DFGPU_AGG_AVG ? 2 : 1, gids.a, nullptr, total));
    }
  }
  // evaluate_group_by + the per-set loop of group_aggregate_batch (aggregates/mod.rs:1161-1200, row_hash.rs:540-600): keys and accumulator
  // arguments are evaluated once per batch; every grouping set interns its own key tuples (masked keys come from null_exprs) into the
  // one GroupValues and updates every accumulator with the resulting group ids.
  void group_aggregate_sets(const TaskContext& tc, Batch& b, const ArrayRef& mask, GroupsRef& groups, std::vector<AccRef>& accs) const {
    std::vector<ArrayRef> keys, nulls, vals(aggs.size()), filts(aggs.size());
    for (auto& e : gexprs) keys.push_back(into_array(tc, e->eval(tc, b), b.base_rows));
    for (auto& e : null_exprs) nulls.push_back(into_array(tc, e->eval(tc, b), b.base_rows));
    for (size_t i = 0; i < aggs.size(); i++) {
      if (merging()) continue;
      if (aggs[i].arg) vals[i] = into_array(tc, aggs[i].arg->eval(tc, b), b.base_rows);
      if (aggs[i].filter) filts[i] = into_array(tc, aggs[i].filter->eval(tc, b), b.base_rows);
    }
    for (auto& set : sets) {
      std::vector<const dfgpu_array*> gp;
      for (size_t i = 0; i < keys.size(); i++) gp.push_back(set[i] ? nulls[i].a : keys[i].a);
      dfgpu_array* ids = nullptr; tc.check(dfgpu_groups_intern(tc.ctx, groups.g, gp.data(), (int32_t)gp.size(), mask.a, &ids)); ArrayRef gids = ArrayRef::adopt(ids);
      int64_t total = dfgpu_groups_len(groups.g);
      size_t col = gexprs.size();
      std::vector<dfgpu_acc*> ap; std::vector<const dfgpu_array*> vp, fp;
      for (size_t i = 0; i < aggs.size(); i++) {
        if (merging()) {
          int nst = aggs[i].kind == DFGPU_AGG_AVG ? 2 : 1; const dfgpu_array* st[2];
          for (int k = 0; k < nst; k++) st[k] = b.column(tc, (int)(col + (size_t)k)).a;
          col += (size_t)nst;
          tc.check(dfgpu_acc_merge_batch(tc.ctx, accs[i].a, st, nst, gids.a, nullptr, total));
        } else { ap.push_back(accs[i].a); vp.push_back(vals[i].a); fp.push_back(filts[i].a); }
      }
      if (!ap.empty()) tc.check(dfgpu_acc_update_batch_multi(tc.ctx, ap.data(), vp.data(), fp.data(), (int32_t)ap.size(), gids.a, total));
    }
  }
  std::vector<std::string> out_names() const { return out_names(mode == 0); }
  std::vector<std::string> out_names(bool as_state) const {
    std::vector<std::string> n = gnames;
    for (auto& a : aggs) { if (as_state) { if (a.kind == DFGPU_AGG_AVG) { n.push_back(a.name + "[count]"); n.push_back(a.name + "[sum]"); } else n.push_back(a.name + "[" + agg_fun_name(a.kind) + "]"); } else n.push_back(a.name); }
    return n;
  }
  SchemaPtr schema() const override { std::lock_guard<std::mutex> l(mu); if (!sch) { auto s = std::make_shared<Schema>(); for (auto& n : out_names()) s->f.push_back(Field{n}); sch = s; } return sch; }
  int partitions() const override { return (mode == 1 || mode == 3) ? 1 : input->partitions(); }
  struct SpillState;
  // ---- GroupedHashAggregateStream (row_hash.rs:423-520): state of one output partition, input pulled batch by batch
  struct AggState {
    TaskContext tc; const ProjectionExec* pj = nullptr; PlanPtr src; std::vector<int> parts; size_t next_part = 0; std::unique_ptr<Stream> cur;
    bool grouped = false, specials = false, input_done = false, emitted_any = false; int64_t fuse_min_rows = 1 << 20, preagg_min_rows = 1 << 22;
    GroupsRef groups; std::vector<AccRef> accs; std::vector<StringMinMax> smm; std::vector<CountDistinct> cds; ArrayRef pending;
    GroupsRef sort_groups; int64_t current_sort = 0;          // GroupOrderingPartial: the sort-key prefixes seen (only the latest is kept), first group of the latest prefix
    std::deque<Batch> ready; std::shared_ptr<SpillState> spill;
    explicit AggState(const TaskContext& t) : tc(t) {}
  };
  std::shared_ptr<AggState> make_state(int partition, const TaskContext& tc) const {
    auto SP = std::make_shared<AggState>(tc); AggState& S = *SP;
    // An input ProjectionExec is looked through: its computed columns that only feed accumulator arguments are evaluated inside the
    // accumulate pass (dfgpu_acc_update_batch_fused) instead of being written out as columns first.
    int64_t fuse_min_rows = 1 << 20; dfgpu_ctx_get_option(tc.ctx, "fused_aggregate_min_rows", &fuse_min_rows);
    int64_t preagg_min_rows = 1 << 22, preagg_on = 1; dfgpu_ctx_get_option(tc.ctx, "agg_partitioned_min_rows", &preagg_min_rows); dfgpu_ctx_get_option(tc.ctx, "agg_partitioned", &preagg_on);
    if (!preagg_on) preagg_min_rows = INT64_MAX;
    const ProjectionExec* pj = (!merging() && !aggs.empty() && fuse_min_rows >= 0) ? dynamic_cast<const ProjectionExec*>(input.get()) : nullptr;
    const PlanPtr& src = pj ? pj->input : input;
    if (mode == 1 || mode == 3) { for (int p = 0; p < src->partitions(); p++) S.parts.push_back(p); } else S.parts.push_back(partition);
    const bool grouped = !gexprs.empty();      // false: AggregateStream (aggregates/no_grouping.rs): one implicit group
    GroupsRef& groups = S.groups; if (grouped) tc.check(dfgpu_groups_new(tc.ctx, (int32_t)gexprs.size(), &groups.g));
    S.accs = std::vector<AccRef>(aggs.size()); S.smm = std::vector<StringMinMax>(aggs.size()); S.cds = std::vector<CountDistinct>(aggs.size());
    std::vector<AccRef>& accs = S.accs; std::vector<StringMinMax>& smm = S.smm; std::vector<CountDistinct>& cds = S.cds;
    const bool specials = any_special();
    for (size_t i = 0; i < aggs.size(); i++) {
      if (is_string_minmax(i)) { smm[i].is_max = aggs[i].kind == DFGPU_AGG_MAX; continue; }
      if (aggs[i].kind == DFGPU_AGG_COUNT_DISTINCT) { cds[i].init(tc); continue; }
      int32_t t = aggs[i].kind == DFGPU_AGG_COUNT ? DFGPU_INT64 : aggs[i].type;
      tc.check(dfgpu_acc_new(tc.ctx, aggs[i].kind, t, aggs[i].precision, aggs[i].scale, &accs[i].a));
    }
    { int64_t lim = 0, ranges = 16; dfgpu_ctx_get_option(tc.ctx, "agg_spill_state_bytes", &lim); dfgpu_ctx_get_option(tc.ctx, "agg_spill_ranges", &ranges);
      if (lim > 0 && grouped && !specials && sets.empty()) { S.spill = std::make_shared<SpillState>(); S.spill->limit = lim; S.spill->ranges = ranges; } }
    if (order_mode != 0) preagg_min_rows = INT64_MAX;        // ordered input is clustered on its keys: run numbering, and the group table has to hold every id for EmitTo::First
    S.pj = pj; S.src = src; S.grouped = grouped; S.specials = specials; S.fuse_min_rows = fuse_min_rows; S.preagg_min_rows = preagg_min_rows;
    return SP;
  }
  void consume(AggState& S, Batch& b_in) const {           // group_aggregate_batch (row_hash.rs:524-613)
    const TaskContext& tc = S.tc; const ProjectionExec* pj = S.pj; const bool grouped = S.grouped, specials = S.specials; const int64_t fuse_min_rows = S.fuse_min_rows, preagg_min_rows = S.preagg_min_rows;
    GroupsRef& groups = S.groups; std::vector<AccRef>& accs = S.accs; std::vector<StringMinMax>& smm = S.smm; std::vector<CountDistinct>& cds = S.cds; ArrayRef& pending = S.pending;
    auto settle_pending = [&]() {  // another batch follows: the keys go into the table after all; first-seen interning of distinct keys numbers them 0 .. n-1 again
      if (!pending) return;
      const dfgpu_array* kp = pending.a; dfgpu_array* ids = nullptr; tc.check(dfgpu_groups_intern(tc.ctx, groups.g, &kp, 1, nullptr, &ids)); ArrayRef drop = ArrayRef::adopt(ids);
      if (dfgpu_groups_len(groups.g) != pending.len()) fail(DFGPU_INTERNAL, "AggregateExec: pre-aggregated keys were not distinct");
      pending = ArrayRef();
    };
    {
      if (b_in.base_rows == 0) return;
      // spill_previous_if_necessary (row_hash.rs:667-683): not in Partial mode, not with an ordered input
      if (S.spill && order_mode == 0 && mode != 0 && !pending && dfgpu_groups_len(groups.g) > 0 && state_bytes(S) > S.spill->limit) spill(S, *S.spill);
      settle_pending();
      Batch raw; std::vector<bool> deferred;
      if (pj) { raw = b_in; b_in = pj->project(tc, raw, &deferred); }
      Batch& b = b_in;
      bool any_deferred = false; for (bool d : deferred) any_deferred |= d;
      auto ensure = [&](int ci) { if (ci >= 0 && ci < (int)deferred.size() && deferred[(size_t)ci]) { pj->evaluate_deferred(tc, raw, b, (size_t)ci); deferred[(size_t)ci] = false; } };
      if (any_deferred) {         // group keys and accumulator filters read their columns the ordinary way
        std::set<int> need; for (auto& e : gexprs) e->columns(need); for (auto& a : aggs) if (a.filter) a.filter->columns(need);
        for (int ci : need) ensure(ci);
      }
      if (b.base_rows == 0) return;
      ArrayRef mask = b.selection; b.selection = ArrayRef();
      if (!sets.empty()) { if (specials) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: COUNT(DISTINCT) / string MIN-MAX under grouping sets on the device"); for (size_t ci = 0; ci < deferred.size(); ci++) ensure((int)ci); group_aggregate_sets(tc, b, mask, groups, accs); return; }
      ArrayRef gids; int64_t total = 1;
      if (grouped) {
        std::vector<ArrayRef> gc; std::vector<const dfgpu_array*> gp;
        for (auto& e : gexprs) {
          // A group key that is still a pending gather take(source, indices) from a small source (a dimension attribute carried through
          // joins: GROUP BY n_name) IS a dictionary array: intern its codes instead of materialising and hashing the values per row.
          int ci = e->column_index();
          if (ci >= 0 && ci < (int)b.cols.size() && !b.cols[(size_t)ci].arr && b.cols[(size_t)ci].source && b.cols[(size_t)ci].source.len() * 4 <= b.base_rows) {
            Col& c = b.cols[(size_t)ci]; dfgpu_array_desc sd; dfgpu_array_describe(c.source.a, &sd);
            if (sd.type != DFGPU_DICTIONARY) {
              dfgpu_array* d = nullptr; tc.check(dfgpu_array_make_dictionary(tc.ctx, col_indices(tc, c).a, c.source.a, &d));
              gc.push_back(ArrayRef::adopt(d)); gp.push_back(gc.back().a); continue;
            }
          }
          gc.push_back(into_array(tc, e->eval(tc, b), b.base_rows)); gp.push_back(gc.back().a);
        }
        // A large batch of high-cardinality keys is first reduced to one row per group partition by partition out of LDS (the Partial stage
        // of a two-phase plan, applied inside the operator): its partial rows are then interned and MERGED like the Final stage does.
        if (!specials && !merging() && gp.size() >= 1 && gp.size() <= 4 && b.base_rows >= preagg_min_rows && preaggregate(tc, pj, raw, b, deferred, ensure, gp, mask, groups, accs, S.spill ? nullptr : &pending)) return;
        dfgpu_array* ids = nullptr; tc.check(specials || order_mode == 1 ? dfgpu_groups_intern(tc.ctx, groups.g, gp.data(), (int32_t)gp.size(), mask.a, &ids) : dfgpu_groups_intern_deferred(tc.ctx, groups.g, gp.data(), (int32_t)gp.size(), mask.a, &ids)); gids = ArrayRef::adopt(ids);      // deferred ids: only the accumulators read them
        total = dfgpu_groups_len(groups.g);
        if (order_mode == 1 && !specials) note_sort_prefix(S, gp, gids, mask);
      } else { dfgpu_array* z = nullptr; tc.check(dfgpu_array_new_zeros(tc.ctx, DFGPU_UINT32, 0, 0, b.base_rows, &z)); gids = ArrayRef::adopt(z); }
      if (!specials && !merging() && total <= 8 && fuse_min_rows >= 0 && b.base_rows >= fuse_min_rows && try_fused(tc, pj, raw, b, deferred, accs, gids, grouped ? ArrayRef() : mask, total)) return;
      for (size_t ci = 0; ci < deferred.size(); ci++) ensure((int)ci);
      size_t col = gexprs.size();
      std::vector<ArrayRef> uvals(aggs.size()), ufilt(aggs.size());       // update mode: all accumulators of the batch go down together
      for (size_t i = 0; i < aggs.size(); i++) {
        if (merging() && is_string_minmax(i)) {           // the state column is the value (min_max.rs state())
          ArrayRef stv = b.column(tc, (int)col); col += 1;
          smm[i].update(tc, gids, stv, mask);              // rows a fused selection dropped carry no state
        } else if (merging() && aggs[i].kind == DFGPU_AGG_COUNT_DISTINCT) {      // the state column is the list of the group's distinct values
          ArrayRef stv = b.column(tc, (int)col); col += 1;
          cds[i].merge(tc, gids, stv, mask, total, aggs[i].type, aggs[i].precision, aggs[i].scale);
        } else if (merging()) {
          int nst = aggs[i].kind == DFGPU_AGG_AVG ? 2 : 1; const dfgpu_array* st[2];
          for (int k = 0; k < nst; k++) st[k] = b.column(tc, (int)(col + (size_t)k)).a;
          col += (size_t)nst;
          tc.check(dfgpu_acc_merge_batch(tc.ctx, accs[i].a, st, nst, gids.a, grouped ? nullptr : mask.a, total));
        } else {
          ArrayRef vals, filt;
          if (aggs[i].arg) vals = into_array(tc, aggs[i].arg->eval(tc, b), b.base_rows);
          if (aggs[i].filter) filt = into_array(tc, aggs[i].filter->eval(tc, b), b.base_rows);
          if (!grouped && mask) {         // rows dropped by a fused FilterExec must not reach the single group
            if (filt) { ArrayRef kf = known_mask(tc, filt); dfgpu_array* o = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_AND, kf.a, 0, mask.a, 0, &o)); filt = ArrayRef::adopt(o); } else filt = mask;
          }
          uvals[i] = vals; ufilt[i] = filt;
        }
      }
      if (!merging() && !aggs.empty()) {
        std::vector<dfgpu_acc*> ap; std::vector<const dfgpu_array*> vp, fp;
        for (size_t i = 0; i < aggs.size(); i++) {
          if (special(i)) {
            ArrayRef f = ufilt[i]; if (grouped && mask) { if (f) { ArrayRef kf = known_mask(tc, f); dfgpu_array* o = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_AND, kf.a, 0, mask.a, 0, &o)); f = ArrayRef::adopt(o); } else f = mask; }
            if (is_string_minmax(i)) smm[i].update(tc, gids, uvals[i], f); else cds[i].update(tc, gids, uvals[i], f, total);
            continue;
          }
          ap.push_back(accs[i].a); vp.push_back(uvals[i].a); fp.push_back(ufilt[i].a);
        }
        if (!ap.empty()) tc.check(dfgpu_acc_update_batch_multi(tc.ctx, ap.data(), vp.data(), fp.data(), (int32_t)ap.size(), gids.a, total));
      }
    }
  }
  // emit(EmitTo::All, spilling) (row_hash.rs:626-662): every group as one batch -- state columns when as_state (Partial output, and what a spill holds), else final values
  bool emit_all(AggState& S, bool as_state, Batch* out) const {
    const TaskContext& tc = S.tc; const bool grouped = S.grouped; GroupsRef& groups = S.groups; std::vector<AccRef>& accs = S.accs; std::vector<StringMinMax>& smm = S.smm; std::vector<CountDistinct>& cds = S.cds; ArrayRef& pending = S.pending;
    int64_t total = grouped ? (pending ? pending.len() : dfgpu_groups_len(groups.g)) : 1;     // no GROUP BY: always one row, even on empty input
    if (total > 0) {              // emit(EmitTo::All) (row_hash.rs:626-662)
      Batch o; o.base_rows = total; std::vector<dfgpu_array*> keys(gexprs.size(), nullptr);
      if (grouped && pending) o.cols.push_back(col_of(pending));          // one key column, already in first-seen order
      else if (grouped) {
        // keys still "column c at the first row of the run" (one clustered batch): they leave as pending gathers, so that a HAVING above reads the keys of the groups it keeps only
        std::vector<dfgpu_array*> src(gexprs.size(), nullptr); dfgpu_array* rows = nullptr;
        dfgpu_status st = dfgpu_groups_emit_deferred(tc.ctx, groups.g, src.data(), &rows);
        if (st == DFGPU_OK) { ArrayRef r = ArrayRef::adopt(rows); MemoPtr memo = std::make_shared<TakeMemo>();
          for (auto sp : src) { Col c; c.source = ArrayRef::adopt(sp); c.chain.push_back(r); c.memo = memo; o.cols.push_back(std::move(c)); } }
        else { if (st != DFGPU_NOT_IMPLEMENTED) tc.check(st); tc.check(dfgpu_groups_emit(tc.ctx, groups.g, keys.data())); for (auto k : keys) o.cols.push_back(col_of(ArrayRef::adopt(k))); }
      }
      dfgpu_array_desc ed{}; ed.type = DFGPU_UINT32; ed.values = &ed; dfgpu_array* e = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &ed, &e)); ArrayRef empty_ids = ArrayRef::adopt(e);
      for (size_t i = 0; i < aggs.size(); i++) {
        if (is_string_minmax(i)) { o.cols.push_back(col_of(smm[i].emit(tc, total))); continue; }          // state and final value are the same column
        if (aggs[i].kind == DFGPU_AGG_COUNT_DISTINCT) { o.cols.push_back(col_of(as_state ? cds[i].state(tc, total, aggs[i].type) : cds[i].emit(tc, total))); continue; }
        tc.check(dfgpu_acc_update_batch(tc.ctx, accs[i].a, nullptr, empty_ids.a, nullptr, total));      // zero-row update: grow the state to `total` groups
        if (as_state) { dfgpu_array* st[2] = {nullptr, nullptr}; int32_t n = 0; tc.check(dfgpu_acc_state(tc.ctx, accs[i].a, st, &n)); for (int k = 0; k < n; k++) o.cols.push_back(col_of(ArrayRef::adopt(st[k]))); }
        else { dfgpu_array* v = nullptr; tc.check(dfgpu_acc_evaluate(tc.ctx, accs[i].a, &v)); o.cols.push_back(col_of(ArrayRef::adopt(v))); }
      }
      auto s = std::make_shared<Schema>(); auto names = out_names(as_state);
      for (size_t i = 0; i < o.cols.size(); i++) s->f.push_back(field_of(names[i], o.cols[i].arr.a));
      o.schema = s; if (as_state == (mode == 0)) { std::lock_guard<std::mutex> l(mu); sch = s; }
      *out = std::move(o); return true;
    }
    return false;
  }
  // ---- spill of the aggregation state to host memory (row_hash.rs:664-771: spill_previous_if_necessary / spill / update_merged_stream).  The reference writes the
  // sorted state to an IPC file per spill and at the end stream-merges the files with the remaining state, re-aggregating the merged, key-ordered stream under
  // GroupOrdering::Full.  Here a spill is the sorted state batch cut into key RANGES (splitter keys fixed at the first spill, so every spill and the remainder are cut at
  // the same keys) and copied to host memory as Arrow arrays; at the end range r of every spill and of the remainder is brought back, merged (merge_batch) in a fresh
  // table, ordered by key and emitted -- ranges in key order, so the output is in key order like the reference's, and the device only ever holds one range's groups.
  struct SpillState { std::vector<SpillRun> runs; std::vector<ArrayRef> splitters; int64_t limit = 0, ranges = 16, spilled_rows = 0, spilled_bytes = 0; };
  int64_t state_bytes(const AggState& S) const { int64_t b = S.groups.g ? dfgpu_groups_size(S.groups.g) : 0; for (auto& a : S.accs) if (a.a) b += dfgpu_acc_size(a.a); return b; }
  void new_accs(AggState& S) const {
    S.accs = std::vector<AccRef>(aggs.size());
    for (size_t i = 0; i < aggs.size(); i++) { if (special(i)) continue; int32_t t = aggs[i].kind == DFGPU_AGG_COUNT ? DFGPU_INT64 : aggs[i].type; S.tc.check(dfgpu_acc_new(S.tc.ctx, aggs[i].kind, t, aggs[i].precision, aggs[i].scale, &S.accs[i].a)); }
  }
  void reset_state(AggState& S) const {            // clear_shrink (row_hash.rs:707-711)
    if (S.groups.g) { dfgpu_groups_free(S.groups.g); S.groups.g = nullptr; } S.tc.check(dfgpu_groups_new(S.tc.ctx, (int32_t)gexprs.size(), &S.groups.g));
    new_accs(S);
  }
  // the state as one batch ordered by the group keys ascending, NULLs first (spill_expr, row_hash.rs:338-345; sort_batch :688), every column materialised
  bool sorted_state(AggState& S, std::vector<ArrayRef>* cols) const {
    Batch b; if (!emit_all(S, true, &b)) return false;
    std::vector<const dfgpu_array*> kp; for (size_t i = 0; i < gexprs.size(); i++) kp.push_back(b.cols[i].arr.a);
    std::vector<uint8_t> desc(kp.size(), 0), nf(kp.size(), 1);
    dfgpu_array* idx = nullptr; S.tc.check(dfgpu_sort_to_indices(S.tc.ctx, kp.data(), desc.data(), nf.data(), (int32_t)kp.size(), -1, &idx)); ArrayRef ix = ArrayRef::adopt(idx);
    cols->clear(); for (auto& c : b.cols) cols->push_back(take(S.tc, c.arr, ix));
    return true;
  }
  // boundaries[r] = first row of range r in a key-sorted state batch (ranges + 1 entries).  Splitter keys and rows are concatenated (splitters first) and ordered by the
  // stable sort: a splitter lands in front of the rows equal to it, so position of splitter j in the order minus j = rows strictly below it.
  std::vector<int64_t> range_bounds(AggState& S, SpillState& P, const std::vector<ArrayRef>& cols) const {
    const TaskContext& tc = S.tc; const int64_t rows = cols[0].len(), K = P.ranges; const size_t nk = gexprs.size();
    if (P.splitters.empty()) {                   // first spill: K - 1 of its rows at equal distances
      std::vector<uint32_t> at; for (int64_t j = 1; j < K; j++) at.push_back((uint32_t)(rows * j / K));
      dfgpu_array_desc d{}; d.type = DFGPU_UINT32; d.length = (int64_t)at.size(); d.values = at.data(); dfgpu_array* ia = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &d, &ia)); ArrayRef ix = ArrayRef::adopt(ia);
      for (size_t k = 0; k < nk; k++) P.splitters.push_back(take(tc, cols[k], ix));
    }
    const int64_t ns = P.splitters[0].len();
    std::vector<ArrayRef> both; std::vector<const dfgpu_array*> bp;
    for (size_t k = 0; k < nk; k++) { const dfgpu_array* two[2] = { P.splitters[k].a, cols[k].a }; dfgpu_array* c = nullptr; tc.check(dfgpu_concat(tc.ctx, two, 2, &c)); both.push_back(ArrayRef::adopt(c)); bp.push_back(both.back().a); }
    std::vector<uint8_t> desc(nk, 0), nf(nk, 1);
    dfgpu_array* idx = nullptr; tc.check(dfgpu_sort_to_indices(tc.ctx, bp.data(), desc.data(), nf.data(), (int32_t)nk, -1, &idx)); ArrayRef order = ArrayRef::adopt(idx);
    uint32_t nsv = (uint32_t)ns; dfgpu_array_desc ld{}; ld.type = DFGPU_UINT32; ld.length = 1; ld.values = &nsv; dfgpu_array* la = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &ld, &la)); ArrayRef lit = ArrayRef::adopt(la);
    dfgpu_array* m = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_LT, order.a, 0, lit.a, 1, &m)); ArrayRef is_split = ArrayRef::adopt(m);
    ArrayRef pos = mask_indices(tc, is_split);
    if (pos.len() != ns) fail(DFGPU_INTERNAL, "AggregateExec spill: %lld splitter positions for %lld splitters", (long long)pos.len(), (long long)ns);
    std::vector<uint32_t> hp((size_t)ns); tc.check(dfgpu_array_export_host(tc.ctx, pos.a, hp.data(), nullptr, nullptr));
    std::vector<int64_t> bounds; bounds.push_back(0);
    for (int64_t j = 0; j < ns; j++) bounds.push_back((int64_t)hp[(size_t)j] - j);      // splitters keep their own order (they are sorted and the sort is stable)
    bounds.push_back(rows);
    return bounds;
  }
  void spill(AggState& S, SpillState& P) const {           // spill (row_hash.rs:685-705)
    std::vector<ArrayRef> cols; if (!sorted_state(S, &cols)) return;
    std::vector<int64_t> bounds = range_bounds(S, P, cols);
    SpillRun run; run.pieces.resize(bounds.size() - 1);
    for (size_t r = 0; r + 1 < bounds.size(); r++) {
      const int64_t lo = bounds[r], len = bounds[r + 1] - bounds[r]; run.pieces[r].rows = len; if (len <= 0) continue;
      for (auto& c : cols) { dfgpu_array* sl = nullptr; S.tc.check(dfgpu_array_slice(S.tc.ctx, c.a, lo, len, &sl)); ArrayRef piece = ArrayRef::adopt(sl);
        run.pieces[r].cols.emplace_back(); HostColumn& h = run.pieces[r].cols.back(); S.tc.check(dfgpu_array_export_arrow(S.tc.ctx, piece.a, &h.a, &h.s)); h.live = true; }
    }
    P.spilled_rows += cols[0].len(); P.runs.push_back(std::move(run));
    cols.clear(); reset_state(S);
  }
  // update_merged_stream + the re-aggregation of the merged stream (row_hash.rs:736-771, :545-600 with is_stream_merging): range by range
  void merge_spills(AggState& S, SpillState& P) const {
    const TaskContext& tc = S.tc; const size_t nk = gexprs.size();
    std::vector<ArrayRef> rest; std::vector<int64_t> rb;
    if (dfgpu_groups_len(S.groups.g) > 0 && sorted_state(S, &rest)) rb = range_bounds(S, P, rest);
    reset_state(S);
    const size_t R = P.runs[0].pieces.size();
    for (size_t r = 0; r < R; r++) {
      auto merge_in = [&](const std::vector<ArrayRef>& cols) {
        std::vector<const dfgpu_array*> kp; for (size_t k = 0; k < nk; k++) kp.push_back(cols[k].a);
        dfgpu_array* ids = nullptr; tc.check(dfgpu_groups_intern(tc.ctx, S.groups.g, kp.data(), (int32_t)nk, nullptr, &ids)); ArrayRef gids = ArrayRef::adopt(ids);
        const int64_t total = dfgpu_groups_len(S.groups.g); size_t col = nk;
        for (size_t i = 0; i < aggs.size(); i++) { const int nst = aggs[i].kind == DFGPU_AGG_AVG ? 2 : 1; const dfgpu_array* st[2]; for (int k = 0; k < nst; k++) st[k] = cols[col + (size_t)k].a; col += (size_t)nst;
          tc.check(dfgpu_acc_merge_batch(tc.ctx, S.accs[i].a, st, nst, gids.a, nullptr, total)); }
      };
      for (auto& run : P.runs) { SpillPiece& pc = run.pieces[r]; if (pc.rows <= 0) continue;
        std::vector<ArrayRef> cols; for (auto& h : pc.cols) { dfgpu_array* a = nullptr; tc.check(dfgpu_array_import_arrow(tc.ctx, &h.a, &h.s, &a)); cols.push_back(ArrayRef::adopt(a)); }
        merge_in(cols); pc.cols.clear(); }
      if (!rest.empty() && rb[r + 1] > rb[r]) { std::vector<ArrayRef> cols; for (auto& c : rest) { dfgpu_array* sl = nullptr; tc.check(dfgpu_array_slice(tc.ctx, c.a, rb[r], rb[r + 1] - rb[r], &sl)); cols.push_back(ArrayRef::adopt(sl)); } merge_in(cols); }
      Batch o; if (!emit_all(S, mode == 0, &o)) { reset_state(S); continue; }
      std::vector<const dfgpu_array*> kp; for (size_t k = 0; k < nk; k++) kp.push_back(o.cols[k].arr.a);
      std::vector<uint8_t> desc(nk, 0), nf(nk, 1);
      dfgpu_array* idx = nullptr; tc.check(dfgpu_sort_to_indices(tc.ctx, kp.data(), desc.data(), nf.data(), (int32_t)nk, -1, &idx)); ArrayRef ix = ArrayRef::adopt(idx);
      for (auto& c : o.cols) c = col_of(take(tc, c.arr, ix));
      S.ready.push_back(std::move(o)); reset_state(S);
    }
    P.runs.clear();
  }
  void finish(AggState& S) const {           // set_input_done_and_produce_output (row_hash.rs:775-790)
    if (S.spill && !S.spill->runs.empty()) { merge_spills(S, *S.spill); return; }
    Batch o; if (emit_all(S, mode == 0, &o)) S.ready.push_back(std::move(o));
  }
  // emit(EmitTo::First(n)) (row_hash.rs:626-662 with groups_accumulator.rs:25-57): the first n groups leave as one batch, the rest are renumbered from 0
  void emit_first(AggState& S, int64_t n) const {
    const TaskContext& tc = S.tc; const int64_t total = dfgpu_groups_len(S.groups.g);
    if (n <= 0 || n > total) return;
    Batch o; o.base_rows = n; std::vector<dfgpu_array*> keys(gexprs.size(), nullptr);
    tc.check(dfgpu_groups_emit_first(tc.ctx, S.groups.g, n, keys.data())); for (auto k : keys) o.cols.push_back(col_of(ArrayRef::adopt(k)));
    dfgpu_array_desc ed{}; ed.type = DFGPU_UINT32; ed.values = &ed; dfgpu_array* e = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &ed, &e)); ArrayRef empty_ids = ArrayRef::adopt(e);
    for (size_t i = 0; i < aggs.size(); i++) {
      tc.check(dfgpu_acc_update_batch(tc.ctx, S.accs[i].a, nullptr, empty_ids.a, nullptr, total));      // zero-row update: grow the state to `total` groups
      dfgpu_array* st[2] = {nullptr, nullptr}; int32_t ns = 0; tc.check(dfgpu_acc_emit_first(tc.ctx, S.accs[i].a, n, mode == 0 ? 1 : 0, st, &ns));
      for (int k = 0; k < ns; k++) o.cols.push_back(col_of(ArrayRef::adopt(st[k])));
    }
    auto sc = std::make_shared<Schema>(); auto names = out_names();
    for (size_t i = 0; i < o.cols.size(); i++) sc->f.push_back(field_of(names[i], o.cols[i].arr.a));
    o.schema = sc; { std::lock_guard<std::mutex> l(mu); sch = sc; }
    S.ready.push_back(std::move(o)); S.emitted_any = true; S.current_sort = S.current_sort > n ? S.current_sort - n : 0;
  }
  // GroupOrderingPartial::new_groups (aggregates/order/partial.rs:196-240): current_sort = group index of the first row that carries the latest sort-key prefix.
  // The prefixes are interned first-seen into their own table (sorted input: equal prefixes are contiguous, so a new prefix is a new id and the latest is the largest);
  // the first row with the largest id, and the group that row fell into, are read back (two 4-byte copies per batch).
  void note_sort_prefix(AggState& S, const std::vector<const dfgpu_array*>& gp, const ArrayRef& gids, const ArrayRef& mask) const {
    const TaskContext& tc = S.tc;
    if (!S.sort_groups.g) tc.check(dfgpu_groups_new(tc.ctx, (int32_t)order_indices.size(), &S.sort_groups.g));
    std::vector<const dfgpu_array*> sk; for (int ix : order_indices) sk.push_back(gp[(size_t)ix]);
    const int64_t before = dfgpu_groups_len(S.sort_groups.g);
    dfgpu_array* si = nullptr; tc.check(dfgpu_groups_intern(tc.ctx, S.sort_groups.g, sk.data(), (int32_t)sk.size(), mask.a, &si)); ArrayRef sids = ArrayRef::adopt(si);
    const int64_t after = dfgpu_groups_len(S.sort_groups.g);
    if (after == before) return;                         // the batch stayed inside the prefix already current
    uint32_t latest = (uint32_t)(after - 1); dfgpu_array_desc ld{}; ld.type = DFGPU_UINT32; ld.length = 1; ld.values = &latest;
    dfgpu_array* la = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &ld, &la)); ArrayRef lit = ArrayRef::adopt(la);
    dfgpu_array* m = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_EQ, sids.a, 0, lit.a, 1, &m)); ArrayRef eq = ArrayRef::adopt(m);
    if (mask) { dfgpu_array* o = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_AND, eq.a, 0, mask.a, 0, &o)); eq = ArrayRef::adopt(o); }
    ArrayRef rows = mask_indices(tc, eq);
    if (rows.len() == 0) fail(DFGPU_INTERNAL, "AggregateExec: no row carries the latest sort prefix");
    dfgpu_array* r0 = nullptr; tc.check(dfgpu_array_slice(tc.ctx, rows.a, 0, 1, &r0)); ArrayRef first_row = ArrayRef::adopt(r0);
    ArrayRef g0 = take(tc, gids, first_row);
    uint32_t gid = 0; tc.check(dfgpu_array_export_host(tc.ctx, g0.a, &gid, nullptr, nullptr));
    if (before == 0 && after == 1) S.current_sort = 0;     // State::Start: the first row's prefix, nothing in front of it
    else S.current_sort = (int64_t)gid;
    if (after > 1) { std::vector<dfgpu_array*> drop(order_indices.size(), nullptr); tc.check(dfgpu_groups_emit_first(tc.ctx, S.sort_groups.g, after - 1, drop.data())); for (auto d : drop) if (d) dfgpu_array_release(d); }
  }
  // GroupOrdering::emit_to after a batch (order/full.rs:87-103, partial.rs:118-132)
  void emit_ordered(AggState& S) const {
    if (order_mode == 0 || !S.grouped || S.specials || !sets.empty() || S.pending) return;
    const int64_t total = dfgpu_groups_len(S.groups.g);
    const int64_t n = order_mode == 2 ? total - 1 : S.current_sort;
    if (n > 0) emit_first(S, n);
  }
  // emit_early_if_necessary (row_hash.rs:720-733): Partial mode over the memory target hands whole batch_size multiples of its groups on and forgets them
  void emit_early(AggState& S) const {
    if (!S.spill || mode != 0 || order_mode != 0 || S.pending) return;
    const int64_t len = dfgpu_groups_len(S.groups.g), bs = S.tc.batch_size > 0 ? S.tc.batch_size : 8192;
    if (len >= bs && state_bytes(S) > S.spill->limit) emit_first(S, len / bs * bs);
  }
  bool pull(AggState& S, Batch& b) const {
    for (;;) {
      if (!S.cur) { if (S.next_part >= S.parts.size()) return false; S.cur = S.src->run(S.parts[S.next_part++], S.tc); }
      if (S.cur->next(b)) return true;
      S.cur.reset();
    }
  }
  struct AggStream : Stream {
    const AggregateExec* op; std::shared_ptr<AggState> st;
    AggStream(const AggregateExec* o, std::shared_ptr<AggState> s) : op(o), st(std::move(s)) {}
    bool next(Batch& out) override {
      AggState& S = *st;
      for (;;) {
        if (!S.ready.empty()) { out = std::move(S.ready.front()); S.ready.pop_front(); return true; }
        if (S.input_done) return false;
        Batch b;
        if (!op->pull(S, b)) { S.input_done = true; op->finish(S); continue; }      // set_input_done_and_produce_output
        op->consume(S, b);
        op->emit_ordered(S);
        op->emit_early(S);
      }
    }
  };
  std::unique_ptr<Stream> execute(int partition, const TaskContext& tc) const override { return std::unique_ptr<Stream>(new AggStream(this, make_state(partition, tc))); }
};

// ------------------------------------------------------------------ SortExec
// sort_batch (sorts/sort.rs:584-609): lexsort_to_indices over the evaluated keys, then take() of every column.  A sort column that is a plain input column may
// come back from the sort already in order (dfgpu_sort_to_indices_keys rebuilds it from the sorted packed keys): that column skips the gather.
static Batch sorted_batch(const TaskContext& tc, Batch& b, const std::vector<ExprPtr>& exprs, const std::vector<const dfgpu_array*>& kp, const std::vector<uint8_t>& desc, const std::vector<uint8_t>& nulls_first, int64_t fetch) {
  // the batch's stored columns that are not sort keys go down as payload (dfgpu_sort_take): a large sort's last pass gathers up to four of them while it writes the result
  std::vector<const dfgpu_array*> pay; std::vector<size_t> pay_col;
  for (size_t ci = 0; ci < b.cols.size(); ci++) {
    bool is_key = false; for (auto& e : exprs) is_key |= e->column_index() == (int)ci;
    if (!is_key && b.cols[ci].arr) { pay.push_back(b.cols[ci].arr.a); pay_col.push_back(ci); }
  }
  dfgpu_array* idx = nullptr; std::vector<dfgpu_array*> sk(kp.size(), nullptr), po(pay.size() + 1, nullptr);
  tc.check(dfgpu_sort_take(tc.ctx, kp.data(), desc.data(), nulls_first.data(), (int32_t)kp.size(), fetch, pay.empty() ? nullptr : pay.data(), (int32_t)pay.size(), &idx, sk.data(), po.data()));
  ArrayRef ix = ArrayRef::adopt(idx); std::vector<ArrayRef> sorted; for (auto* a : sk) sorted.push_back(a ? ArrayRef::adopt(a) : ArrayRef());
  std::vector<ArrayRef> taken(b.cols.size()); for (size_t j = 0; j < pay.size(); j++) if (po[j]) taken[pay_col[j]] = ArrayRef::adopt(po[j]);
  Batch o; o.schema = b.schema; o.base_rows = ix.len();
  MemoPtr memo = std::make_shared<TakeMemo>();
  for (size_t ci = 0; ci < b.cols.size(); ci++) {
    ArrayRef ready; for (size_t e = 0; e < exprs.size() && !ready; e++) if (sorted[e] && exprs[e]->column_index() == (int)ci) ready = sorted[e];
    if (!ready && taken[ci]) ready = taken[ci];
    o.cols.push_back(ready ? col_of(ready) : col_take(b.cols[ci], ix, memo));
  }
  return o;
}

struct SortExec : Plan {          // sorts/sort.rs:719-733; sort_batch :584-609
  std::vector<ExprPtr> exprs; std::vector<uint8_t> desc, nulls_first; int64_t fetch; bool preserve; PlanPtr input;
  PlanPtr fresh() const override { auto s = std::make_shared<SortExec>(); s->exprs = exprs; s->desc = desc; s->nulls_first = nulls_first; s->fetch = fetch; s->preserve = preserve; s->input = input->fresh(); return s; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {input}; }
  const char* name() const override { return "SortExec"; }
  SchemaPtr schema() const override { return input->schema(); }
  int partitions() const override { return preserve ? input->partitions() : 1; }
  // ---- spill (ExternalSorter, sorts/sort.rs:208-400).  The reference keeps the input batches in memory until the reservation fails, then sorts them and writes the sorted run
  // to an IPC file; at the end the runs and the batches still in memory go through the streaming merge.  Here, with the context option "sort_spill_bytes" > 0 (and no fetch:
  // a TopK keeps `fetch` rows, it never spills), the batches are kept until they exceed that many bytes; then they are sorted, the sorted run is cut at SPLITTER keys
  // ("sort_spill_ranges" - 1 rows of the first run at equal distances, the same keys for every later run) and the pieces are copied to host memory as Arrow arrays.  At the
  // end range r of every run is brought back, concatenated in run order and sorted once more (stable: equal keys keep their arrival order, as in the in-memory sort); the
  // ranges come out in key order, one batch each, so the result is the sorted input and the device never holds more than the budget plus one range.
  struct Spilled : Stream {
    const SortExec* op; TaskContext tc; std::vector<SpillRun> runs; size_t r = 0;
    Spilled(const SortExec* o, TaskContext t, std::vector<SpillRun> rs) : op(o), tc(t), runs(std::move(rs)) {}
    bool next(Batch& out) override {
      const size_t R = runs.empty() ? 0 : runs[0].pieces.size();
      for (; r < R; r++) {
        std::vector<Batch> parts;
        for (auto& run : runs) { SpillPiece& pc = run.pieces[r]; if (pc.rows <= 0) continue;
          Batch b; b.schema = op->schema(); b.base_rows = pc.rows;
          for (auto& h : pc.cols) { dfgpu_array* a = nullptr; tc.check(dfgpu_array_import_arrow(tc.ctx, &h.a, &h.s, &a)); b.cols.push_back(col_of(ArrayRef::adopt(a))); }
          pc.cols.clear(); parts.push_back(std::move(b)); }
        Batch all; if (parts.empty() || !concat_batches(tc, parts, &all) || all.base_rows <= 0) continue;
        parts.clear();
        std::vector<ArrayRef> keys; std::vector<const dfgpu_array*> kp;
        for (auto& e : op->exprs) { keys.push_back(into_array(tc, e->eval(tc, all), all.base_rows)); kp.push_back(keys.back().a); }
        out = sorted_batch(tc, all, op->exprs, kp, op->desc, op->nulls_first, -1); r++; return true;
      }
      return false;
    }
  };
  // first row of every range in a batch sorted on `keys` (ranges + 1 entries): splitters and rows are concatenated, splitters first, and ranked by the stable sort -- a splitter
  // lands in front of the rows equal to it, so (its rank - its own number) rows are strictly before it
  std::vector<int64_t> range_bounds(const TaskContext& tc, std::vector<ArrayRef>& splitters, const std::vector<ArrayRef>& keys, int64_t K) const {
    const int64_t rows = keys[0].len(); const size_t nk = keys.size();
    if (splitters.empty()) {
      std::vector<uint32_t> at; for (int64_t j = 1; j < K; j++) at.push_back((uint32_t)(rows * j / K));
      if (at.empty()) at.push_back(0);               // one range: a splitter in front of everything keeps the code below uniform (range 0 is empty, range 1 is the run)
      dfgpu_array_desc d{}; d.type = DFGPU_UINT32; d.length = (int64_t)at.size(); d.values = at.data(); dfgpu_array* ia = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &d, &ia)); ArrayRef ix = ArrayRef::adopt(ia);
      for (size_t k = 0; k < nk; k++) splitters.push_back(take(tc, keys[k], ix));
    }
    const int64_t ns = splitters[0].len();
    std::vector<ArrayRef> both; std::vector<const dfgpu_array*> bp;
    for (size_t k = 0; k < nk; k++) { const dfgpu_array* two[2] = { splitters[k].a, keys[k].a }; dfgpu_array* c = nullptr; tc.check(dfgpu_concat(tc.ctx, two, 2, &c)); both.push_back(ArrayRef::adopt(c)); bp.push_back(both.back().a); }
    dfgpu_array* idx = nullptr; tc.check(dfgpu_sort_to_indices(tc.ctx, bp.data(), desc.data(), nulls_first.data(), (int32_t)nk, -1, &idx)); ArrayRef order = ArrayRef::adopt(idx);
    uint32_t nsv = (uint32_t)ns; dfgpu_array_desc ld{}; ld.type = DFGPU_UINT32; ld.length = 1; ld.values = &nsv; dfgpu_array* la = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &ld, &la)); ArrayRef lit = ArrayRef::adopt(la);
    dfgpu_array* m = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_LT, order.a, 0, lit.a, 1, &m)); ArrayRef is_split = ArrayRef::adopt(m);
    ArrayRef pos = mask_indices(tc, is_split);
    if (pos.len() != ns) fail(DFGPU_INTERNAL, "SortExec spill: %lld splitter positions for %lld splitters", (long long)pos.len(), (long long)ns);
    std::vector<uint32_t> hp((size_t)ns); tc.check(dfgpu_array_export_host(tc.ctx, pos.a, hp.data(), nullptr, nullptr));
    std::vector<int64_t> bounds; bounds.push_back(0);
    for (int64_t j = 0; j < ns; j++) bounds.push_back((int64_t)hp[(size_t)j] - j);      // the splitters are sorted among themselves (rows of a sorted run) and the sort is stable
    bounds.push_back(rows);
    return bounds;
  }
  static int64_t batch_bytes(const Batch& b) {
    int64_t t = 0;
    for (auto& c : b.cols) { dfgpu_array_desc d; if (!c.arr.a) continue; dfgpu_array_describe(c.arr.a, &d);
      const int64_t w = d.type == DFGPU_UTF8 ? 4 : d.type == DFGPU_DICTIONARY ? 4 : d.type == DFGPU_BOOL ? 1 : d.type == DFGPU_DECIMAL128 ? 16 : d.type == DFGPU_INT32 || d.type == DFGPU_UINT32 || d.type == DFGPU_DATE32 || d.type == DFGPU_FLOAT32 ? 4 : d.type == DFGPU_INT8 || d.type == DFGPU_UINT8 ? 1 : d.type == DFGPU_INT16 || d.type == DFGPU_UINT16 ? 2 : 8;
      t += d.length * w + (d.type == DFGPU_UTF8 ? d.values_bytes : 0) + (d.validity ? (d.length + 7) / 8 : 0); }
    return t;
  }
  void spill_run(const TaskContext& tc, std::vector<Batch>& mem, std::vector<SpillRun>& runs, std::vector<ArrayRef>& splitters, int64_t K) const {
    Batch all; if (mem.empty() || !concat_batches(tc, mem, &all) || all.base_rows <= 0) { mem.clear(); return; }
    mem.clear();
    Batch sorted;
    { std::vector<ArrayRef> keys; std::vector<const dfgpu_array*> kp;
      for (auto& e : exprs) { keys.push_back(into_array(tc, e->eval(tc, all), all.base_rows)); kp.push_back(keys.back().a); }
      sorted = sorted_batch(tc, all, exprs, kp, desc, nulls_first, -1); all = Batch(); }
    std::vector<ArrayRef> cols; for (auto& c : sorted.cols) cols.push_back(col_get(tc, c));
    Batch view; view.schema = sorted.schema; view.base_rows = sorted.base_rows; for (auto& c : cols) view.cols.push_back(col_of(c));
    std::vector<ArrayRef> skeys; for (auto& e : exprs) skeys.push_back(into_array(tc, e->eval(tc, view), view.base_rows));
    std::vector<int64_t> bounds = range_bounds(tc, splitters, skeys, K);
    SpillRun run; run.pieces.resize(bounds.size() - 1); int64_t bytes = 0;
    for (size_t r = 0; r + 1 < bounds.size(); r++) {
      const int64_t lo = bounds[r], len = bounds[r + 1] - bounds[r]; run.pieces[r].rows = len; if (len <= 0) continue;
      for (auto& c : cols) { dfgpu_array* sl = nullptr; tc.check(dfgpu_array_slice(tc.ctx, c.a, lo, len, &sl)); ArrayRef piece = ArrayRef::adopt(sl);
        run.pieces[r].cols.emplace_back(); HostColumn& h = run.pieces[r].cols.back(); tc.check(dfgpu_array_export_arrow(tc.ctx, piece.a, &h.a, &h.s)); h.live = true; }
    }
    bytes = batch_bytes(view);
    { std::lock_guard<std::mutex> l(met->mu); met->spill_count++; met->spilled_rows += view.base_rows; met->spilled_bytes += bytes; }
    runs.push_back(std::move(run));
  }
  std::unique_ptr<Stream> execute(int partition, const TaskContext& tc) const override {
    int64_t budget = 0, K = 16; dfgpu_ctx_get_option(tc.ctx, "sort_spill_bytes", &budget); dfgpu_ctx_get_option(tc.ctx, "sort_spill_ranges", &K);
    std::vector<Batch> in;
    if (budget > 0 && fetch < 0 && !exprs.empty()) {
      std::vector<SpillRun> runs; std::vector<ArrayRef> splitters; int64_t held = 0;
      auto pull = [&](int p) {
        auto s = input->run(p, tc); Batch b;
        while (s->next(b)) { Batch m = materialize(tc, b); for (auto& c : m.cols) c = col_of(col_get(tc, c)); held += batch_bytes(m); in.push_back(std::move(m)); b = Batch();
          if (held > budget) { spill_run(tc, in, runs, splitters, K); held = 0; } }
      };
      if (preserve) pull(partition); else for (int p = 0; p < input->partitions(); p++) pull(p);
      if (!runs.empty()) {                               // what is still in memory becomes the last run (sort.rs:283-313 merges it with the spills the same way)
        spill_run(tc, in, runs, splitters, K);
        return std::unique_ptr<Stream>(new Spilled(this, tc, std::move(runs)));
      }
    } else {
      if (preserve) drain(input, partition, tc, in); else for (int p = 0; p < input->partitions(); p++) drain(input, p, tc, in);
    }
    std::vector<Batch> outv; Batch b;
    if (!in.empty() && concat_batches(tc, in, &b) && b.base_rows > 0) {
      std::vector<ArrayRef> keys; std::vector<const dfgpu_array*> kp;
      for (auto& e : exprs) { keys.push_back(into_array(tc, e->eval(tc, b), b.base_rows)); kp.push_back(keys.back().a); }
      outv.push_back(sorted_batch(tc, b, exprs, kp, desc, nulls_first, fetch));
    }
    return std::unique_ptr<Stream>(new VecStream(std::move(outv)));
  }
};

// ------------------------------------------------------------------ SortPreservingMergeExec
// sorts/sort_preserving_merge.rs:67-120, execute :186-247; streaming_merge / loser tree (sorts/merge.rs:38-110) picks, among streams whose heads compare equal, the lower
// stream index, and holds one batch per input.  On the device a merge step works on whole CHUNKS: every input is loaded up to its share of "spm_merge_rows" rows (a ctx option,
// 2^25 by default); the loaded rows of all inputs are concatenated in input order and ranked by the stable radix sort of dfgpu_sort_to_indices, which for sorted inputs IS the
// k-way merge order, ties included.  If every input ended inside its chunk that is the whole answer, in one pass (the usual in-memory case).  Otherwise only the rows that
// no unloaded row can precede may leave: the bound is the smallest (last loaded key, input index) among the inputs that have more to deliver; a copy of that row is put into
// the concatenation right behind its own input, so the stable sort ranks it behind every loaded row that precedes the first unloaded one and in front of all others -- the rows
// ranked before it are emitted, the rest of each input (a suffix: the inputs are sorted) stays loaded, and the input that set the bound, now empty, loads its next chunk.
struct SortPreservingMergeExec : Plan {
  std::vector<ExprPtr> exprs; std::vector<uint8_t> desc, nulls_first; int64_t fetch; PlanPtr input;
  PlanPtr fresh() const override { auto s = std::make_shared<SortPreservingMergeExec>(); s->exprs = exprs; s->desc = desc; s->nulls_first = nulls_first; s->fetch = fetch; s->input = input->fresh(); return s; }
  std::vector<std::shared_ptr<const Plan>> children() const override { return {input}; }
  const char* name() const override { return "SortPreservingMergeExec"; }
  SchemaPtr schema() const override { return input->schema(); }
  int partitions() const override { return 1; }
  struct Merge : Stream {
    const SortPreservingMergeExec* op; TaskContext tc;
    struct In { std::unique_ptr<Stream> s; std::vector<Batch> loaded; int64_t rows = 0; Batch ahead; bool has_ahead = false, done = false; };
    std::vector<In> in; int64_t share = 1, emitted = 0; bool finished = false;
    Merge(const SortPreservingMergeExec* o, TaskContext t) : op(o), tc(t) {
      const int np = o->input->partitions(); int64_t budget = 0; dfgpu_ctx_get_option(tc.ctx, "spm_merge_rows", &budget); if (budget <= 0) budget = (int64_t)1 << 25;
      share = std::max<int64_t>(1, budget / np);
      in.resize((size_t)np); for (int p = 0; p < np; p++) in[(size_t)p].s = o->input->run(p, tc);
    }
    // one batch ahead, so that an input whose last batch has been loaded is known to be done (≙ the merge's cursors: `is_finished`)
    bool pull(In& x, Batch* b) {
      if (x.has_ahead) { *b = std::move(x.ahead); x.has_ahead = false; return true; }
      if (x.done) return false;
      Batch t; while (x.s->next(t)) { Batch m = materialize(tc, t); if (m.base_rows > 0) { *b = std::move(m); return true; } }
      x.done = true; return false;
    }
    void fill(In& x) {
      Batch b;
      while (x.rows < share && pull(x, &b)) { x.rows += b.base_rows; x.loaded.push_back(std::move(b)); }
      if (!x.done && !x.has_ahead) { Batch n; if (pull(x, &n)) { x.ahead = std::move(n); x.has_ahead = true; } }
    }
    bool more(const In& x) const { return x.has_ahead || !x.done; }
    Batch one(In& x) { Batch o; if (x.loaded.size() == 1) o = std::move(x.loaded[0]); else concat_batches(tc, x.loaded, &o); x.loaded.clear(); for (auto& c : o.cols) c = col_of(col_get(tc, c)); return o; }
    static Batch rows_of(const TaskContext& tc, const Batch& b, int64_t off, int64_t len) {
      Batch o; o.schema = b.schema; o.base_rows = len;
      for (auto& c : b.cols) { dfgpu_array* sl = nullptr; tc.check(dfgpu_array_slice(tc.ctx, c.arr.a, off, len, &sl)); o.cols.push_back(col_of(ArrayRef::adopt(sl))); }
      return o;
    }
    ArrayRef order_of(Batch& all, int64_t limit) {
      std::vector<ArrayRef> keys; std::vector<const dfgpu_array*> kp;
      for (auto& e : op->exprs) { keys.push_back(into_array(tc, e->eval(tc, all), all.base_rows)); kp.push_back(keys.back().a); }
      dfgpu_array* idx = nullptr; tc.check(dfgpu_sort_to_indices(tc.ctx, kp.data(), op->desc.data(), op->nulls_first.data(), (int32_t)kp.size(), limit, &idx)); return ArrayRef::adopt(idx);
    }
    Batch gather(const Batch& all, const ArrayRef& ix) { Batch o; o.schema = all.schema; o.base_rows = ix.len(); for (auto& c : all.cols) o.cols.push_back(col_of(take(tc, c.arr, ix))); return o; }
    bool next(Batch& out) override {
      while (!finished) {
        for (auto& x : in) if (x.rows == 0) fill(x);
        std::vector<size_t> act; for (size_t p = 0; p < in.size(); p++) if (in[p].rows > 0) act.push_back(p);
        if (act.empty()) { finished = true; break; }
        const int64_t left = op->fetch >= 0 ? op->fetch - emitted : -1;
        if (left == 0) { finished = true; break; }
        std::vector<Batch> parts; std::vector<int64_t> off; int64_t total = 0;
        for (size_t p : act) { parts.push_back(one(in[p])); in[p].loaded.clear(); off.push_back(total); total += parts.back().base_rows; }
        bool bounded = false; for (size_t p : act) bounded |= more(in[p]);
        if (!bounded) {                                    // every input ended inside its chunk: what is loaded is all there is
          Batch all; if (parts.size() == 1) all = std::move(parts[0]); else concat_batches(tc, parts, &all);
          for (size_t p : act) in[p].rows = 0;
          ArrayRef ix = order_of(all, left);
          out = gather(all, ix); emitted += out.base_rows; finished = true;
          if (out.base_rows > 0) return true;
          break;
        }
        // the bound: the smallest (last loaded row, input index) among the inputs with more to come -- ranked by the same stable sort over those rows, in input order
        size_t bi = 0;
        { std::vector<Batch> lasts; std::vector<size_t> who;
          for (size_t k = 0; k < act.size(); k++) if (more(in[act[k]])) { lasts.push_back(rows_of(tc, parts[k], parts[k].base_rows - 1, 1)); who.push_back(k); }
          if (who.size() == 1) bi = who[0];
          else { Batch lb; concat_batches(tc, lasts, &lb); ArrayRef o1 = order_of(lb, 1); uint32_t first = 0; tc.check(dfgpu_array_export_host(tc.ctx, o1.a, &first, nullptr, nullptr)); bi = who.at(first); } }
        // concatenation: inputs in order, the bound row's copy right behind its own input
        std::vector<Batch> seq; int64_t sentinel = -1, at = 0; std::vector<int64_t> start(act.size());
        for (size_t k = 0; k < act.size(); k++) { start[k] = at; at += parts[k].base_rows; seq.push_back(parts[k]); if (k == bi) { sentinel = at; at += 1; seq.push_back(rows_of(tc, parts[k], parts[k].base_rows - 1, 1)); } }
        Batch all; concat_batches(tc, seq, &all); seq.clear();
        ArrayRef ix = order_of(all, -1);
        // where the copy landed = how many rows may leave
        uint32_t sv = (uint32_t)sentinel; dfgpu_array_desc ld{}; ld.type = DFGPU_UINT32; ld.length = 1; ld.values = &sv; dfgpu_array* la = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &ld, &la)); ArrayRef lit = ArrayRef::adopt(la);
        dfgpu_array* m = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_EQ, ix.a, 0, lit.a, 1, &m)); ArrayRef is_s = ArrayRef::adopt(m);
        ArrayRef where = mask_indices(tc, is_s); if (where.len() != 1) fail(DFGPU_INTERNAL, "SortPreservingMergeExec: the bound row was ranked %lld times", (long long)where.len());
        uint32_t cut = 0; tc.check(dfgpu_array_export_host(tc.ctx, where.a, &cut, nullptr, nullptr));
        dfgpu_array* hd = nullptr; tc.check(dfgpu_array_slice(tc.ctx, ix.a, 0, (int64_t)cut, &hd)); ArrayRef head = ArrayRef::adopt(hd);
        // rows leaving per input: an input's emitted rows are a prefix of it (it is sorted); the prefix length is the number of its rows ranked before the cut
        std::vector<int64_t> gone(act.size(), 0);
        { const dfgpu_array* pk = ix.a; uint8_t no = 0; dfgpu_array* rk = nullptr; tc.check(dfgpu_sort_to_indices(tc.ctx, &pk, &no, &no, 1, -1, &rk)); ArrayRef rank = ArrayRef::adopt(rk);      // argsort of a permutation = its inverse: rank[i] = position of row i
          uint32_t cv = cut; dfgpu_array_desc cd{}; cd.type = DFGPU_UINT32; cd.length = 1; cd.values = &cv; dfgpu_array* ca = nullptr; tc.check(dfgpu_array_import_host(tc.ctx, &cd, &ca)); ArrayRef clit = ArrayRef::adopt(ca);
          dfgpu_array* bm = nullptr; tc.check(dfgpu_binary(tc.ctx, DFGPU_OP_LT, rank.a, 0, clit.a, 1, &bm)); ArrayRef before = ArrayRef::adopt(bm);
          for (size_t k = 0; k < act.size(); k++) { dfgpu_array* sl = nullptr; tc.check(dfgpu_array_slice(tc.ctx, before.a, start[k], parts[k].base_rows, &sl)); ArrayRef part = ArrayRef::adopt(sl); tc.check(dfgpu_mask_count(tc.ctx, part.a, &gone[k])); } }
        Batch o = gather(all, head);
        for (size_t k = 0; k < act.size(); k++) {
          In& x = in[act[k]]; const int64_t keep = parts[k].base_rows - gone[k];
          if (k == bi && keep != 0) fail(DFGPU_INTERNAL, "SortPreservingMergeExec: %lld rows of the bounding input stayed behind", (long long)keep);
          x.rows = keep; if (keep > 0) x.loaded.push_back(rows_of(tc, parts[k], gone[k], keep));
        }
        if (left >= 0 && o.base_rows > left) { o = rows_of(tc, o, 0, left); }
        emitted += o.base_rows;
        if (o.base_rows > 0) { out = std::move(o); return true; }
      }
      return false;
    }
  };
  std::unique_ptr<Stream> execute(int partition, const TaskContext& tc) const override {
    if (partition != 0) fail(DFGPU_INTERNAL, "SortPreservingMergeExec invalid partition %d", partition);
    int np = input->partitions();
    if (np == 0) fail(DFGPU_INTERNAL, "SortPreservingMergeExec requires at least one input partition");
    if (np == 1) return input->run(0, tc);              // bypass (:213-218)
    if (exprs.empty()) fail(DFGPU_INTERNAL, "Sort expressions cannot be empty for streaming merge");      // sorts/streaming_merge.rs
    return std::unique_ptr<Stream>(new Merge(this, tc));
  }
};

}  // namespace dfx

// ==================================================================== C ABI
using namespace dfx;
struct dfgpu_expr { ExprPtr e; };
struct dfgpu_plan { PlanPtr p; };
struct dfgpu_batch { Batch b; };
struct dfgpu_stream { std::unique_ptr<Stream> s; PlanPtr keep; TaskContext tc; };

template <typename F> static dfgpu_status guard(F&& f) {
  try { f(); return DFGPU_OK; }
  catch (const Err& e) { g_err = e.msg; return e.code; }
  catch (const std::bad_alloc&) { g_err = "host allocation failed"; return DFGPU_RESOURCES_EXHAUSTED; }
  catch (const std::exception& e) { g_err = e.what(); return DFGPU_INTERNAL; }
}
static ExprPtr ex(const dfgpu_expr* e) { if (!e) fail(DFGPU_INVALID_ARGUMENT, "null expression"); return e->e; }
static PlanPtr pl(const dfgpu_plan* p) { if (!p) fail(DFGPU_INVALID_ARGUMENT, "null plan"); return p->p; }

extern "C" {

const char* dfgpu_exec_last_error(void) { return g_err.c_str(); }

dfgpu_status dfgpu_batch_new(const char* const* names, const dfgpu_array* const* columns, int32_t ncols, dfgpu_batch** out) {
  return guard([&] {
    if (ncols < 0 || !out) fail(DFGPU_INVALID_ARGUMENT, "batch_new: bad arguments");
    auto* b = new dfgpu_batch(); b->b.schema = std::make_shared<Schema>();
    for (int i = 0; i < ncols; i++) {
      if (!columns[i]) { delete b; fail(DFGPU_INVALID_ARGUMENT, "batch_new: null column"); }
      if (i && dfgpu_array_length(columns[i]) != dfgpu_array_length(columns[0])) { delete b; fail(DFGPU_INVALID_ARGUMENT, "batch_new: columns differ in length"); }
      b->b.cols.push_back(col_of(ArrayRef::share(columns[i]))); b->b.schema->f.push_back(field_of(names && names[i] ? names[i] : "", columns[i]));
    }
    b->b.base_rows = ncols ? dfgpu_array_length(columns[0]) : 0;
    *out = b;
  });
}
void dfgpu_batch_free(dfgpu_batch* b) { delete b; }
int32_t dfgpu_batch_num_columns(const dfgpu_batch* b) { return b ? (int32_t)b->b.cols.size() : 0; }
const char* dfgpu_batch_column_name(const dfgpu_batch* b, int32_t i) { return (b && i >= 0 && i < (int)b->b.schema->f.size()) ? b->b.schema->f[(size_t)i].name.c_str() : ""; }
dfgpu_status dfgpu_batch_num_rows(dfgpu_ctx* ctx, dfgpu_batch* b, int64_t* out) { return guard([&] { if (!ctx || !b || !out) fail(DFGPU_INVALID_ARGUMENT, "batch_num_rows: null argument"); TaskContext tc{ctx, 8192}; *out = num_rows(tc, b->b); }); }
// every pending gather of the batch: columns that go through the same index array are gathered together (dfgpu_take_multi)
static void materialize_all(const TaskContext& tc, Batch& b) {
  if (b.selection) b = materialize(tc, b);
  std::map<const dfgpu_array*, std::vector<size_t>> groups;
  for (size_t i = 0; i < b.cols.size(); i++) { Col& c = b.cols[i]; if (!c.arr && c.source && !c.chain.empty()) groups[col_indices(tc, c).a].push_back(i); }
  for (auto& g : groups) {
    if (g.second.size() < 2) continue;
    std::vector<const dfgpu_array*> vals; for (size_t i : g.second) vals.push_back(b.cols[i].source.a);
    std::vector<dfgpu_array*> outs(vals.size(), nullptr);
    ArrayRef idx = b.cols[g.second[0]].chain[0];
    tc.check(dfgpu_take_multi(tc.ctx, vals.data(), (int32_t)vals.size(), idx.a, outs.data()));
    for (size_t k = 0; k < g.second.size(); k++) { Col& c = b.cols[g.second[k]]; c.arr = ArrayRef::adopt(outs[k]); c.source = ArrayRef(); c.chain.clear(); c.memo.reset(); }
  }
  for (auto& c : b.cols) (void)col_get(tc, c);
}
dfgpu_status dfgpu_batch_materialize(dfgpu_ctx* ctx, dfgpu_batch* b) {
  return guard([&] { if (!ctx || !b) fail(DFGPU_INVALID_ARGUMENT, "batch_materialize: null argument"); TaskContext tc{ctx, 8192}; materialize_all(tc, b->b); });
}
dfgpu_status dfgpu_batch_column(dfgpu_ctx* ctx, dfgpu_batch* b, int32_t i, dfgpu_array** out) {
  return guard([&] {
    if (!ctx || !b || !out) fail(DFGPU_INVALID_ARGUMENT, "batch_column: null argument");
    TaskContext tc{ctx, 8192};
    if (b->b.selection) b->b = materialize(tc, b->b);
    const ArrayRef& a = b->b.column(tc, i); dfgpu_array_retain(a.a); *out = a.a;
  });
}

dfgpu_status dfgpu_expr_column(const char* name, int32_t index, dfgpu_expr** out) { return guard([&] { if (!out) fail(DFGPU_INVALID_ARGUMENT, "expr_column: null argument"); *out = new dfgpu_expr{std::make_shared<ColumnExpr>(name ? name : "", index)}; }); }
dfgpu_status dfgpu_expr_literal(const dfgpu_array* s, dfgpu_expr** out) {
  return guard([&] { if (!s || dfgpu_array_length(s) != 1) fail(DFGPU_INVALID_ARGUMENT, "literal must be a length-1 array"); *out = new dfgpu_expr{std::make_shared<LiteralExpr>(ArrayRef::share(s))}; });
}
dfgpu_status dfgpu_expr_binary(const dfgpu_expr* l, int32_t op, const dfgpu_expr* r, dfgpu_expr** out) { return guard([&] { *out = new dfgpu_expr{std::make_shared<BinaryExpr>(ex(l), op, ex(r))}; }); }
static dfgpu_status unary(const dfgpu_expr* e, int kind, int a0, int a1, int a2, const dfgpu_array* list, dfgpu_expr** out) {
  return guard([&] { auto u = std::make_shared<UnaryExpr>(); u->e = ex(e); u->kind = kind; u->a0 = a0; u->a1 = a1; u->a2 = a2; if (list) u->list = ArrayRef::share(list); *out = new dfgpu_expr{u}; });
}
dfgpu_status dfgpu_expr_not(const dfgpu_expr* e, dfgpu_expr** out) { return unary(e, 0, 0, 0, 0, nullptr, out); }
dfgpu_status dfgpu_expr_is_null(const dfgpu_expr* e, int32_t negated, dfgpu_expr** out) { return unary(e, negated ? 2 : 1, 0, 0, 0, nullptr, out); }
dfgpu_status dfgpu_expr_negative(const dfgpu_expr* e, dfgpu_expr** out) { return unary(e, 3, 0, 0, 0, nullptr, out); }
dfgpu_status dfgpu_expr_cast(const dfgpu_expr* e, int32_t t, int32_t p, int32_t s, dfgpu_expr** out) { return unary(e, 4, t, p, s, nullptr, out); }
dfgpu_status dfgpu_expr_in_list(const dfgpu_expr* e, const dfgpu_array* list, int32_t negated, dfgpu_expr** out) { return unary(e, negated ? 6 : 5, 0, 0, 0, list, out); }
void dfgpu_expr_free(dfgpu_expr* e) { delete e; }

dfgpu_status dfgpu_plan_memory(const dfgpu_batch* const* batches, const int32_t* sizes, int32_t nparts, dfgpu_plan** out) {
  return guard([&] {
    if (!out || nparts < 0 || (nparts > 0 && (!sizes || !batches))) fail(DFGPU_INVALID_ARGUMENT, "plan_memory: null argument");
    auto m = std::make_shared<MemoryExec>(); int k = 0;
    for (int p = 0; p < nparts; p++) { m->parts.emplace_back(); for (int i = 0; i < sizes[p]; i++) { const dfgpu_batch* b = batches[k++]; if (!b) fail(DFGPU_INVALID_ARGUMENT, "null batch"); if (!m->sch) m->sch = b->b.schema; m->parts.back().push_back(b->b); } }
    if (!m->sch) m->sch = std::make_shared<Schema>();
    *out = new dfgpu_plan{m};
  });
}
dfgpu_status dfgpu_plan_parquet(dfgpu_parquet* file, const int32_t* columns, int32_t ncols, int32_t npartitions, int32_t row_groups_per_batch, dfgpu_plan** out) {
  return guard([&] {
    if (!file || !out || (ncols > 0 && !columns) || npartitions < 1) fail(DFGPU_INVALID_ARGUMENT, "plan_parquet: bad argument");
    auto n = std::make_shared<ParquetExec>(); n->file = file; n->nparts = npartitions; n->per_batch = row_groups_per_batch > 0 ? row_groups_per_batch : 1; n->sch = std::make_shared<Schema>();
    for (int32_t i = 0; i < ncols; i++) {
      int32_t t = 0, vt = 0, pr = 0, sc = 0, nl = 0;
      if (dfgpu_parquet_column_type(file, columns[i], &t, &vt, &pr, &sc, &nl) != DFGPU_OK) fail(DFGPU_INVALID_ARGUMENT, "plan_parquet: column %d is not in the file", columns[i]);
      if (!t) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: Parquet column '%s' has a type outside the device scan", dfgpu_parquet_column_name(file, columns[i]));
      n->proj.push_back(columns[i]); n->sch->f.push_back(Field{dfgpu_parquet_column_name(file, columns[i]), vt, pr, sc});
    }
    *out = new dfgpu_plan{n};
  });
}
dfgpu_status dfgpu_plan_csv(const uint8_t* bytes, int64_t len, int32_t delimiter, int32_t quote, int32_t escape, int32_t has_header, const char* const* names, const int32_t* types, int32_t ncols_file,
                            const int32_t* columns, int32_t ncols, int32_t npartitions, int64_t batch_bytes, dfgpu_plan** out) {
  return guard([&] {
    if ((!bytes && len) || len < 0 || !out || !names || !types || ncols_file < 1 || ncols < 1 || !columns || npartitions < 1) fail(DFGPU_INVALID_ARGUMENT, "plan_csv: bad argument");
    auto n = std::make_shared<CsvExec>(); n->bytes = bytes; n->len = len; n->delim = delimiter; n->quote = quote; n->escape = escape; n->header = has_header != 0; n->ncols_file = ncols_file; n->nparts = npartitions;
    n->sch = std::make_shared<Schema>();
    for (int32_t i = 0; i < ncols; i++) {
      const int32_t c = columns[i];
      if (c < 0 || c >= ncols_file || (i && c <= columns[i - 1])) fail(DFGPU_INVALID_ARGUMENT, "plan_csv: projected columns must be ascending file column indices");
      n->proj.push_back(c); n->types.insert(n->types.end(), {types[3 * c], types[3 * c + 1], types[3 * c + 2]});
      n->sch->f.push_back(Field{names[c] ? names[c] : "", types[3 * c], types[3 * c + 1], types[3 * c + 2]});
    }
    n->cut(batch_bytes > 0 ? batch_bytes : (int64_t)256 << 20);
    *out = new dfgpu_plan{n};
  });
}
dfgpu_status dfgpu_plan_parquet_prune(dfgpu_plan* p, int32_t column, int64_t min_value, int64_t max_value) {
  return guard([&] {
    auto* n = p ? dynamic_cast<const ParquetExec*>(p->p.get()) : nullptr;
    if (!n) fail(DFGPU_INVALID_ARGUMENT, "plan_parquet_prune: not a ParquetExec");
    std::lock_guard<std::mutex> l(n->mu); n->bounds.push_back(ParquetExec::Bound{column, min_value, max_value});
  });
}
int64_t dfgpu_plan_parquet_pruned(const dfgpu_plan* p) { auto* n = p ? dynamic_cast<const ParquetExec*>(p->p.get()) : nullptr; return n ? n->pruned.load() : -1; }
dfgpu_status dfgpu_plan_memory_replace(dfgpu_plan* p, const dfgpu_batch* const* batches, const int32_t* sizes, int32_t nparts) {
  return guard([&] {
    auto* m = p ? dynamic_cast<const MemoryExec*>(p->p.get()) : nullptr;
    if (!m) fail(DFGPU_INVALID_ARGUMENT, "plan_memory_replace: not a MemoryExec");
    std::vector<std::vector<Batch>> np; int k = 0;
    for (int i = 0; i < nparts; i++) { np.emplace_back(); for (int j = 0; j < sizes[i]; j++) { const dfgpu_batch* b = batches[k++]; if (!b) fail(DFGPU_INVALID_ARGUMENT, "null batch"); np.back().push_back(b->b); } }
    std::lock_guard<std::mutex> l(m->mu); m->parts = std::move(np);
  });
}
dfgpu_status dfgpu_plan_filter(const dfgpu_expr* pred, const dfgpu_plan* input, dfgpu_plan** out) { return guard([&] { auto f = std::make_shared<FilterExec>(); f->pred = ex(pred); f->input = pl(input); *out = new dfgpu_plan{f}; }); }
dfgpu_status dfgpu_plan_projection(const dfgpu_expr* const* exprs, const char* const* names, int32_t n, const dfgpu_plan* input, dfgpu_plan** out) {
  return guard([&] { auto p = std::make_shared<ProjectionExec>(); for (int i = 0; i < n; i++) { p->exprs.push_back(ex(exprs[i])); p->names.push_back(names[i] ? names[i] : ""); } p->input = pl(input); *out = new dfgpu_plan{p}; });
}
dfgpu_status dfgpu_plan_coalesce_batches(const dfgpu_plan* input, int64_t target, dfgpu_plan** out) { return guard([&] { auto c = std::make_shared<CoalesceBatchesExec>(); c->input = pl(input); c->target = target; *out = new dfgpu_plan{c}; }); }
dfgpu_status dfgpu_plan_coalesce_partitions(const dfgpu_plan* input, dfgpu_plan** out) { return guard([&] { auto c = std::make_shared<CoalescePartitionsExec>(); c->input = pl(input); *out = new dfgpu_plan{c}; }); }
// the nearest HashJoinExec below `p` through operators that pass a selection on untouched (CoalesceBatchesExec, a ProjectionExec of plain columns) learns that its
// consumer fuses selections
static void mark_selection_consumer(const PlanPtr& p) {
  const Plan* q = p.get();
  while (q) {
    if (auto* hj = dynamic_cast<const HashJoinExec*>(q)) { hj->selection_consumer = true; return; }
    if (auto* cb = dynamic_cast<const CoalesceBatchesExec*>(q)) q = cb->input.get();
    else if (auto* pr = dynamic_cast<const ProjectionExec*>(q)) { if (!pr->only_columns()) return; q = pr->input.get(); }
    else return;
  }
}
dfgpu_status dfgpu_plan_repartition(const dfgpu_plan* input, const dfgpu_expr* const* exprs, int32_t nexprs, int32_t n, dfgpu_plan** out) {
  return guard([&] { if (n < 1) fail(DFGPU_INVALID_ARGUMENT, "repartition: partition count must be positive"); auto r = std::make_shared<RepartitionExec>(); r->input = pl(input); r->n = n; if (nexprs > 0) mark_selection_consumer(r->input); for (int i = 0; i < nexprs; i++) r->exprs.push_back(ex(exprs[i])); *out = new dfgpu_plan{r}; });
}
dfgpu_status dfgpu_plan_hash_join(const dfgpu_plan* left, const dfgpu_plan* right, const dfgpu_expr* const* on_left, const dfgpu_expr* const* on_right, int32_t non,
                                  const dfgpu_expr* filter, const int32_t* fs, const int32_t* fi, int32_t nf, int32_t join_type, int32_t mode, int32_t nen, dfgpu_plan** out) {
  return guard([&] {
    if (!left || !right || !out) fail(DFGPU_INVALID_ARGUMENT, "plan_hash_join: null argument");
    if (non < 1) fail(DFGPU_EXECUTION, "Plan error: On constraints in HashJoinExec should be non-empty");       // hash_join.rs:303-305
    if (join_type < 0 || join_type > DFGPU_JOIN_RIGHT_ANTI) fail(DFGPU_INVALID_ARGUMENT, "unknown join type %d", join_type);
    auto j = std::make_shared<HashJoinExec>(); j->left = pl(left); j->right = pl(right);
    for (int i = 0; i < non; i++) { j->on_l.push_back(ex(on_left[i])); j->on_r.push_back(ex(on_right[i])); }
    if (filter) { j->filter = ex(filter); for (int i = 0; i < nf; i++) { j->f_side.push_back(fs[i]); j->f_index.push_back(fi[i]); } }
    j->join_type = join_type; j->mode = mode; j->null_equals_null = nen != 0;
    mark_selection_consumer(j->left); mark_selection_consumer(j->right);
    *out = new dfgpu_plan{j};
  });
}
dfgpu_status dfgpu_plan_sort_merge_join(const dfgpu_plan* left, const dfgpu_plan* right, const dfgpu_expr* const* on_left, const dfgpu_expr* const* on_right, int32_t non, const dfgpu_expr* filter,
                                        const int32_t* fs, const int32_t* fi, int32_t nf, int32_t join_type, int32_t null_equals_null, dfgpu_plan** out) {
  return guard([&] {
    if (!left || !right || !out) fail(DFGPU_INVALID_ARGUMENT, "plan_sort_merge_join: null argument");
    if (non < 1) fail(DFGPU_EXECUTION, "Plan error: On constraints in SortMergeJoinExec should be non-empty");       // sort_merge_join.rs:116-120
    if (join_type == DFGPU_JOIN_RIGHT_SEMI) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: SortMergeJoinExec does not support JoinType::RightSemi");       // :107-111
    if (join_type < 0 || join_type > DFGPU_JOIN_RIGHT_ANTI) fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: SortMergeJoinExec join type %d on the device", join_type);
    if (filter && (join_type == DFGPU_JOIN_LEFT_SEMI || join_type == DFGPU_JOIN_LEFT_ANTI || join_type == DFGPU_JOIN_RIGHT_ANTI))
      fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: SortMergeJoinExec %s with a JoinFilter on the device (Inner, Left, Right and Full take one)", join_type == DFGPU_JOIN_LEFT_SEMI ? "LeftSemi" : join_type == DFGPU_JOIN_LEFT_ANTI ? "LeftAnti" : "RightAnti");
    if (filter && nf > 0 && (!fs || !fi)) fail(DFGPU_INVALID_ARGUMENT, "sort_merge_join: JoinFilter without column indices");
    auto j = std::make_shared<SortMergeJoinExec>(); j->left = pl(left); j->right = pl(right);
    if (filter) { j->filter = ex(filter); for (int i = 0; i < nf; i++) { j->f_side.push_back(fs[i]); j->f_index.push_back(fi[i]); } }
    for (int i = 0; i < non; i++) { j->on_l.push_back(ex(on_left[i])); j->on_r.push_back(ex(on_right[i])); }
    j->join_type = join_type; j->null_equals_null = null_equals_null != 0;
    *out = new dfgpu_plan{j};
  });
}
dfgpu_status dfgpu_plan_nested_loop_join(const dfgpu_plan* left, const dfgpu_plan* right, const dfgpu_expr* filter, const int32_t* fs, const int32_t* fi, int32_t nf, int32_t join_type, dfgpu_plan** out) {
  return guard([&] {
    if (join_type < 0 || join_type > DFGPU_JOIN_RIGHT_ANTI) fail(DFGPU_INVALID_ARGUMENT, "unknown join type %d", join_type);
    auto j = std::make_shared<NestedLoopJoinExec>(); j->left = pl(left); j->right = pl(right); j->join_type = join_type;
    if (filter) { j->filter = ex(filter); for (int i = 0; i < nf; i++) { j->f_side.push_back(fs[i]); j->f_index.push_back(fi[i]); } }
    *out = new dfgpu_plan{j};
  });
}
dfgpu_status dfgpu_plan_aggregate(int32_t mode, const dfgpu_expr* const* gexprs, const char* const* gnames, int32_t ng, const int32_t* kinds, const dfgpu_expr* const* args,
                                  const dfgpu_expr* const* filters, const char* const* names, const int32_t* types, int32_t na, const dfgpu_plan* input, dfgpu_plan** out) {
  return guard([&] {
    if (mode < 0 || mode > 4) fail(DFGPU_INVALID_ARGUMENT, "unknown AggregateMode %d", mode);
    auto a = std::make_shared<AggregateExec>(); a->mode = mode; a->input = pl(input);
    for (int i = 0; i < ng; i++) { a->gexprs.push_back(ex(gexprs[i])); a->gnames.push_back(gnames[i] ? gnames[i] : ""); }
    for (int i = 0; i < na; i++) {
      AggExpr x; x.kind = kinds[i]; if (args && args[i]) x.arg = args[i]->e; if (filters && filters[i]) x.filter = filters[i]->e; x.name = names[i] ? names[i] : "";
      x.type = types[3 * i]; x.precision = types[3 * i + 1]; x.scale = types[3 * i + 2];
      if (x.kind < DFGPU_AGG_SUM || x.kind > DFGPU_AGG_COUNT_DISTINCT) fail(DFGPU_NOT_IMPLEMENTED, "aggregate kind %d has no GroupsAccumulator on device", x.kind);
      if (x.kind == DFGPU_AGG_COUNT_DISTINCT && mode != 3 && mode != 4 && (x.type == DFGPU_DICTIONARY || x.type == DFGPU_BOOL))
        fail(DFGPU_NOT_IMPLEMENTED, "This feature is not implemented: COUNT(DISTINCT) over a type %d argument in AggregateMode %d on the device (its Partial state is a List per group, carried for fixed-width and Utf8 values); Single / SinglePartitioned are", x.type, mode);
      if (!x.arg && x.kind != DFGPU_AGG_COUNT && mode != 1 && mode != 2) fail(DFGPU_INVALID_ARGUMENT, "aggregate %s needs an argument", x.name.c_str());
      a->aggs.push_back(std::move(x));
    }
    *out = new dfgpu_plan{a};
  });
}
dfgpu_status dfgpu_plan_aggregate_input_order(dfgpu_plan* aggregate, int32_t input_order_mode, const int32_t* order_indices, int32_t n) {
  return guard([&] {
    auto* a = aggregate ? const_cast<AggregateExec*>(dynamic_cast<const AggregateExec*>(aggregate->p.get())) : nullptr;       // the node is still private to its builder
    if (!a) fail(DFGPU_INVALID_ARGUMENT, "plan_aggregate_input_order: not an AggregateExec");
    if (input_order_mode < 0 || input_order_mode > 2 || (input_order_mode == 1 && (n < 1 || !order_indices))) fail(DFGPU_INVALID_ARGUMENT, "plan_aggregate_input_order: mode %d with %d order indices", input_order_mode, n);
    a->order_indices.clear();
    if (input_order_mode == 1) for (int i = 0; i < n; i++) { if (order_indices[i] < 0 || order_indices[i] >= (int32_t)a->gexprs.size()) fail(DFGPU_INVALID_ARGUMENT, "plan_aggregate_input_order: order index %d of %zu group expressions", order_indices[i], a->gexprs.size()); a->order_indices.push_back(order_indices[i]); }
    a->order_mode = input_order_mode;
  });
}
dfgpu_status dfgpu_plan_aggregate_grouping_sets(dfgpu_plan* aggregate, const dfgpu_expr* const* null_exprs, int32_t nkeys, const uint8_t* groups, int32_t nsets) {
  return guard([&] {
    auto* a = aggregate ? const_cast<AggregateExec*>(dynamic_cast<const AggregateExec*>(aggregate->p.get())) : nullptr;       // the node is still private to its builder
    if (!a) fail(DFGPU_INVALID_ARGUMENT, "plan_aggregate_grouping_sets: not an AggregateExec");
    if (nkeys != (int32_t)a->gexprs.size() || nkeys < 1 || nsets < 1 || !null_exprs || !groups) fail(DFGPU_INVALID_ARGUMENT, "plan_aggregate_grouping_sets: %d null expressions / %d sets for %zu group expressions", nkeys, nsets, a->gexprs.size());
    a->null_exprs.clear(); a->sets.clear();
    for (int i = 0; i < nkeys; i++) a->null_exprs.push_back(ex(null_exprs[i]));
    for (int s2 = 0; s2 < nsets; s2++) a->sets.emplace_back(groups + (size_t)s2 * (size_t)nkeys, groups + (size_t)(s2 + 1) * (size_t)nkeys);
  });
}
// A SortExec whose keys hold every group column of the AggregateExec below it (reached through CoalesceBatchesExec, FilterExec and ProjectionExecs of plain columns, which
// keep or drop rows and rename columns but never reorder them): group key tuples are distinct, so the sort keys order the rows totally and the order in which the
// aggregation emits its groups cannot show.  The aggregation is told (any_group_order).
static const AggregateExec* any_group_order_target(const SortExec& s) {
  std::set<int> cols; for (auto& e : s.exprs) { const int ci = e->column_index(); if (ci >= 0) cols.insert(ci); }
  const Plan* q = s.input.get();
  while (q) {
    if (auto* a = dynamic_cast<const AggregateExec*>(q)) {
      if (a->mode == 0 || !a->sets.empty() || a->order_mode != 0 || a->gexprs.empty()) return nullptr;          // partial states, grouping sets, ordered streaming: left alone
      for (size_t g = 0; g < a->gexprs.size(); g++) if (!cols.count((int)g)) return nullptr;
      return a;
    }
    if (auto* cb = dynamic_cast<const CoalesceBatchesExec*>(q)) q = cb->input.get();
    else if (auto* f = dynamic_cast<const FilterExec*>(q)) q = f->input.get();
    else if (auto* pr = dynamic_cast<const ProjectionExec*>(q)) {
      if (!pr->only_columns()) return nullptr;
      std::set<int> below; for (int ci : cols) { if (ci >= (int)pr->exprs.size()) return nullptr; below.insert(pr->exprs[(size_t)ci]->column_index()); }
      cols = below; q = pr->input.get();
    } else return nullptr;
  }
  return nullptr;
}
// The mark goes on the sort's OWN copy of its input (fresh(): the same plan with new operator state): the caller's handle on the aggregation, executed on its own, keeps
// emitting in first-seen order.
static void mark_any_group_order(SortExec& s) {
  if (!any_group_order_target(s)) return;
  s.input = s.input->fresh();
  if (auto* a = any_group_order_target(s)) a->any_group_order = true;
}
dfgpu_status dfgpu_plan_sort(const dfgpu_expr* const* exprs, const uint8_t* desc, const uint8_t* nf, int32_t n, int64_t fetch, int32_t preserve, const dfgpu_plan* input, dfgpu_plan** out) {
  return guard([&] {
    if (n < 1) fail(DFGPU_INVALID_ARGUMENT, "Sort requires at least one column");
    auto s = std::make_shared<SortExec>(); s->input = pl(input); s->fetch = fetch; s->preserve = preserve != 0;
    for (int i = 0; i < n; i++) { s->exprs.push_back(ex(exprs[i])); s->desc.push_back(desc ? desc[i] : 0); s->nulls_first.push_back(nf ? nf[i] : 1); }
    mark_any_group_order(*s);
    *out = new dfgpu_plan{s};
  });
}
dfgpu_status dfgpu_plan_sort_preserving_merge(const dfgpu_expr* const* exprs, const uint8_t* desc, const uint8_t* nf, int32_t n, int64_t fetch, const dfgpu_plan* input, dfgpu_plan** out) {
  return guard([&] {
    auto s = std::make_shared<SortPreservingMergeExec>(); s->input = pl(input); s->fetch = fetch;
    for (int i = 0; i < n; i++) { s->exprs.push_back(ex(exprs[i])); s->desc.push_back(desc ? desc[i] : 0); s->nulls_first.push_back(nf ? nf[i] : 1); }
    *out = new dfgpu_plan{s};
  });
}
void dfgpu_plan_free(dfgpu_plan* p) { delete p; }
dfgpu_status dfgpu_plan_with_fresh_state(const dfgpu_plan* p, dfgpu_plan** out) { return guard([&] { if (!out) fail(DFGPU_INVALID_ARGUMENT, "plan_with_fresh_state: null argument"); *out = new dfgpu_plan{pl(p)->fresh()}; }); }
int32_t dfgpu_plan_partition_count(const dfgpu_plan* p) { return p ? p->p->partitions() : 0; }
int32_t dfgpu_plan_schema_len(const dfgpu_plan* p) { if (!p) return 0; auto s = p->p->schema(); return s ? (int32_t)s->f.size() : 0; }
const char* dfgpu_plan_schema_name(const dfgpu_plan* p, int32_t i) {
  static thread_local std::string name; if (!p) return ""; auto s = p->p->schema(); if (!s || i < 0 || i >= (int)s->f.size()) return ""; name = s->f[(size_t)i].name; return name.c_str();
}
const char* dfgpu_plan_name(const dfgpu_plan* p) { return p ? p->p->name() : ""; }

// One C-ABI call that drives operators = one deferred region for kernel error flags (include/dfgpu.h "defer_flag_checks"):
// the flags are read back once, when the region is left, and the call returns that error -- before any batch is handed out.
struct FlagRegion {
  dfgpu_ctx* c; bool open = true;
  explicit FlagRegion(dfgpu_ctx* c_) : c(c_) { dfgpu_ctx_set_option(c, "defer_flag_checks", 1); }
  dfgpu_status close() { open = false; return dfgpu_ctx_set_option(c, "defer_flag_checks", 0); }
  ~FlagRegion() { if (open) close(); }
};
static void metrics_lines(const std::shared_ptr<const Plan>& p, int depth, std::string& out, int64_t* inclusive_out) {
  p->met->resolve();
  int64_t child_ns = 0; std::string below;
  for (auto& c : p->children()) { int64_t ci = 0; metrics_lines(c, depth + 1, below, &ci); child_ns += ci; }
  Metrics& m = *p->met; std::lock_guard<std::mutex> l(m.mu);
  const int64_t inclusive = m.ns[0] + m.ns[1];           // a CollectLeft build runs inside the first poll of ONE stream; count it once
  int64_t self = m.ns[0] - child_ns; if (self < 0) self = 0;
  char line[512]; int k = snprintf(line, sizeof line, "%d %s output_rows=%lld output_batches=%lld elapsed_compute=%lld", depth, p->name(), (long long)m.output_rows, (long long)m.output_batches, (long long)self);
  if (!strcmp(p->name(), "HashJoinExec")) k += snprintf(line + k, sizeof line - (size_t)k, " build_time=%lld join_time=%lld", (long long)m.ns[1], (long long)m.ns[2]);
  if (!strcmp(p->name(), "RepartitionExec")) k += snprintf(line + k, sizeof line - (size_t)k, " repartition_time=%lld", (long long)m.ns[3]);
  if (!strcmp(p->name(), "SortExec")) k += snprintf(line + k, sizeof line - (size_t)k, " spill_count=%lld spilled_bytes=%lld spilled_rows=%lld", (long long)m.spill_count, (long long)m.spilled_bytes, (long long)m.spilled_rows);
  out += line; out += "\n"; out += below;
  if (inclusive_out) *inclusive_out = m.ns[0];
  (void)inclusive;
}
/* see include/dfgpu_exec.h */
dfgpu_status dfgpu_plan_metrics(const dfgpu_plan* p, char* buf, int64_t capacity) {
  return guard([&] {
    if (!p || !buf || capacity < 1) fail(DFGPU_INVALID_ARGUMENT, "plan_metrics: null argument");
    std::string out; metrics_lines(p->p, 0, out, nullptr);
    if ((int64_t)out.size() + 1 > capacity) fail(DFGPU_INVALID_ARGUMENT, "plan_metrics: buffer of %lld bytes, need %zu", (long long)capacity, out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
  });
}
dfgpu_status dfgpu_plan_execute(const dfgpu_plan* p, int32_t partition, dfgpu_ctx* ctx, int64_t batch_size, dfgpu_stream** out) {
  return guard([&] {
    if (!p || !ctx || !out) fail(DFGPU_INVALID_ARGUMENT, "plan_execute: null argument");
    TaskContext tc{ctx, batch_size > 0 ? batch_size : 8192};
    { int64_t m = 0; dfgpu_ctx_get_option(ctx, "collect_metrics", &m); tc.metrics = m != 0; }
    auto* s = new dfgpu_stream{nullptr, p->p, tc};
    FlagRegion region(ctx);           // operators that drain their input when the stream is created (SortExec, build sides)
    try { s->s = p->p->run(partition, tc); tc.check(region.close()); } catch (...) { delete s; throw; }
    *out = s;
  });
}
dfgpu_status dfgpu_stream_next(dfgpu_stream* s, dfgpu_batch** out) {
  return guard([&] {
    if (!s || !out) fail(DFGPU_INVALID_ARGUMENT, "stream_next: null argument");
    FlagRegion region(s->tc.ctx);
    Batch b; bool more = s->s->next(b);
    s->tc.check(region.close());
    if (!more) { *out = nullptr; return; }
    *out = new dfgpu_batch{std::move(b)};
  });
}
void dfgpu_stream_free(dfgpu_stream* s) { delete s; }

}  // extern "C"
