// abi_cbor.hip — extern "C" entry points of include/relearn_hip.h, part: actor serialisation (host side only; kernels live in kernels_*.hip).
#include "abi_internal.hpp"

extern "C" {

// ---------------------------------------------------------------- actor serialisation (serde_cbor layout)
static void cbor_interval(cbor::Writer &w, double lo, double hi) {
  w.map(2);  // IntervalSpace { low, high } (spaces/interval.rs:14-18)
  w.key("low");
  w.f64(lo);
  w.key("high");
  w.f64(hi);
}

static void cbor_tensor(cbor::Writer &w, const float *data, std::initializer_list<int64_t> shape) {
  w.map(5);  // TensorDef (torch/serialize.rs:62-81)
  w.key("kind");
  w.text("Float");
  w.key("shape");
  w.array(shape.size());
  size_t count = 1;
  for (int64_t d : shape) {
    w.sint(d);
    count *= (size_t)d;
  }
  w.key("requires_grad");
  w.boolean(true);
  w.key("byte_order");
  w.text("LittleEndian");
  w.key("data");
  w.bytes(data, count * sizeof(float));
}

// Mlp { layers, activation, output_activation } over `p` = [W1, b1, W2, b2] (ff/mlp.rs:45-50, ff/linear.rs:43-50)
static void cbor_mlp(cbor::Writer &w, const float *p, int64_t in, int64_t hid, int64_t out) {
  w.map(3);
  w.key("layers");
  w.array(2);
  const int64_t dims[2][2] = {{in, hid}, {hid, out}};
  for (int l = 0; l < 2; ++l) {
    w.map(2);
    w.key("kernel");
    cbor_tensor(w, p, {dims[l][1], dims[l][0]});
    p += dims[l][0] * dims[l][1];
    w.key("bias");
    cbor_tensor(w, p, {dims[l][1]});
    p += dims[l][1];
  }
  w.key("activation");
  w.text("Relu");
  w.key("output_activation");
  w.text("Identity");
}

static void cbor_module(cbor::Writer &w, const rl_mlp *m, const std::vector<float> &p) {
  if (m->kind == RL_MODULE_MLP) {
    cbor_mlp(w, p.data(), m->in_dim, m->hidden, m->out_dim);
    return;
  }
  // Gru and Lstm are both RnnBase<impl> (seq/rnn/gru.rs:17, lstm.rs:12): the same document, gate rows 3H or 4H
  const int64_t H = m->gru_hidden, D = m->in_dim, GHR = (int64_t)rl_module_gates(m->kind) * H;
  w.map(3);  // Chain { first, second, activation } (modules/chain.rs:58-63)
  w.key("first");
  w.map(4);  // RnnBase { weights, hidden_size, dropout, type_ } (`device` is #[serde(skip)], seq/rnn/mod.rs:90-99)
  w.key("weights");
  w.map(2);  // RnnWeights { flat_weights, has_biases } (seq/rnn/mod.rs:186-191)
  w.key("flat_weights");
  w.array(4);
  const float *q = p.data();
  cbor_tensor(w, q, {GHR, D});
  q += GHR * D;
  cbor_tensor(w, q, {GHR, H});
  q += GHR * H;
  cbor_tensor(w, q, {GHR});
  q += GHR;
  cbor_tensor(w, q, {GHR});
  q += GHR;
  w.key("has_biases");
  w.boolean(true);
  w.key("hidden_size");
  w.uint((uint64_t)H);
  w.key("dropout");
  w.f64(0.0);
  w.key("type_");
  w.null();  // PhantomData
  w.key("second");
  cbor_mlp(w, q, H, m->hidden, m->out_dim);
  w.key("activation");
  w.text("Relu");
}

static void cbor_observation_space(cbor::Writer &w, const rl_env *env) {
  w.map(1);  // NonEmptyFeatures { inner } (spaces/nonempty_features.rs:20-25)
  w.key("inner");
  auto inner = [&]() {
    if (env->kind != RL_ENV_CARTPOLE) {
      w.map(1);  // IndexSpace { size } (spaces/index.rs:19-22)
      w.key("size");
      w.uint(env->dev.chain_size);
      return;
    }
    // CartPolePhysicalStateSpace (envs/cartpole.rs:73-82, 273-284); default intervals = [f64::MIN, f64::MAX]
    const double lo = -1.7976931348623157e308, hi = 1.7976931348623157e308;
    w.map(4);
    w.key("cart_position");
    cbor_interval(w, -env->cfg.cartpole.max_pos, env->cfg.cartpole.max_pos);
    w.key("cart_velocity");
    cbor_interval(w, lo, hi);
    w.key("pole_angle");
    cbor_interval(w, -env->cfg.cartpole.max_angle, env->cfg.cartpole.max_angle);
    w.key("pole_angular_velocity");
    cbor_interval(w, lo, hi);
  };
  if (env->cfg.limit_kind == RL_LIMIT_VISIBLE) {
    w.map(2);  // StepLimitObsSpace { inner, remaining } (wrappers/step_limit.rs:133-138)
    w.key("inner");
    inner();
    w.key("remaining");
    cbor_interval(w, 0.0, 1.0);
  } else {
    inner();  // the latent limit and the bare env keep the env's own observation space
  }
}

int32_t rl_actor_to_cbor(rl_env *env, rl_mlp *module, int32_t actor_kind, double exploration_rate, uint8_t *buf,
                         uint64_t cap, uint64_t *len_out) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && module && len_out, "NULL argument");
    RL_REQUIRE(actor_kind == RL_ACTOR_POLICY || actor_kind == RL_ACTOR_DQN, "unknown actor kind");
    RL_REQUIRE(module->eng == env->eng, "handles belong to different engines");
    RL_REQUIRE(module->in_dim == env->D && module->out_dim == env->A, "module shape does not match the env");
    std::vector<float> p(module->P);
    d2h(module->eng, p.data(), module->d_params, module->P * sizeof(float));
    cbor::Writer w;
    w.map(actor_kind == RL_ACTOR_DQN ? 4 : 3);
    w.key("observation_space");
    cbor_observation_space(w, env);
    w.key("action_space");
    w.map(0);  // IndexedTypeSpace<T>: its only field is #[serde(skip)] (spaces/indexed_type.rs:57-64)
    w.key(actor_kind == RL_ACTOR_DQN ? "action_value_fn" : "policy_module");
    cbor_module(w, module, p);
    if (actor_kind == RL_ACTOR_DQN) {
      w.key("exploration_rate");
      w.f64(exploration_rate);
    }
    *len_out = w.out.size();
    if (buf != nullptr) {
      RL_REQUIRE(cap >= w.out.size(), "buffer too small for the CBOR document");
      std::memcpy(buf, w.out.data(), w.out.size());
    }
  });
}

static void cbor_read_tensor(const cbor::Value &t, std::initializer_list<int64_t> shape, float *dst) {
  RL_REQUIRE(t.at("kind").s == "Float", "CBOR tensor: kind must be Float");
  RL_REQUIRE(t.at("byte_order").s == "LittleEndian", "CBOR tensor: data has non-native byte order");
  const cbor::Value &sh = t.at("shape");
  RL_REQUIRE(sh.kind == cbor::Value::ARRAY && sh.items.size() == shape.size(), "CBOR tensor: unexpected rank");
  size_t count = 1, i = 0;
  for (int64_t d : shape) {
    RL_REQUIRE(sh.items[i++]->as_int() == d, "CBOR tensor: unexpected shape");
    count *= (size_t)d;
  }
  const cbor::Value &data = t.at("data");
  RL_REQUIRE(data.kind == cbor::Value::BYTES && data.s.size() == count * sizeof(float), "CBOR tensor: bad data length");
  std::memcpy(dst, data.s.data(), data.s.size());
}

static float *cbor_read_mlp(const cbor::Value &m, int64_t in, int64_t hid, int64_t out, float *dst) {
  RL_REQUIRE(m.at("activation").s == "Relu" && m.at("output_activation").s == "Identity",
             "CBOR module: only Relu hidden / Identity output activations are built");
  const cbor::Value &layers = m.at("layers");
  RL_REQUIRE(layers.kind == cbor::Value::ARRAY && layers.items.size() == 2, "CBOR module: expected one hidden layer");
  const int64_t dims[2][2] = {{in, hid}, {hid, out}};
  for (int l = 0; l < 2; ++l) {
    const cbor::Value &lin = *layers.items[l];
    cbor_read_tensor(lin.at("kernel"), {dims[l][1], dims[l][0]}, dst);
    dst += dims[l][0] * dims[l][1];
    RL_REQUIRE(lin.at("bias").kind == cbor::Value::MAP, "CBOR module: layers without bias are not built");
    cbor_read_tensor(lin.at("bias"), {dims[l][1]}, dst);
    dst += dims[l][1];
  }
  return dst;
}

int32_t rl_module_from_cbor(rl_mlp *module, const uint8_t *buf, uint64_t len) {
  return guarded(module ? module->eng : nullptr, [&] {
    RL_REQUIRE(module && buf, "NULL argument");
    cbor::ValuePtr doc = cbor::Reader(buf, (size_t)len).parse();
    const cbor::Value &mod = doc->has("policy_module") ? doc->at("policy_module") : doc->at("action_value_fn");
    std::vector<float> p(module->P);
    float *end;
    if (module->kind == RL_MODULE_MLP) {
      end = cbor_read_mlp(mod, module->in_dim, module->hidden, module->out_dim, p.data());
    } else {
      const int64_t H = module->gru_hidden, D = module->in_dim, GHR = (int64_t)rl_module_gates(module->kind) * H;
      RL_REQUIRE(mod.at("activation").s == "Relu", "CBOR module: Chain activation must be Relu");
      const cbor::Value &rnn = mod.at("first");
      RL_REQUIRE(rnn.at("hidden_size").as_int() == H, "CBOR module: GRU hidden size mismatch");
      RL_REQUIRE(rnn.at("dropout").as_float() == 0.0, "CBOR module: dropout is not built");
      const cbor::Value &wts = rnn.at("weights");
      RL_REQUIRE(wts.at("has_biases").kind == cbor::Value::BOOL && wts.at("has_biases").b, "CBOR module: GRU biases required");
      const cbor::Value &fw = wts.at("flat_weights");
      RL_REQUIRE(fw.kind == cbor::Value::ARRAY && fw.items.size() == 4, "CBOR module: expected a one-layer GRU");
      float *q = p.data();
      cbor_read_tensor(*fw.items[0], {GHR, D}, q);
      q += GHR * D;
      cbor_read_tensor(*fw.items[1], {GHR, H}, q);
      q += GHR * H;
      cbor_read_tensor(*fw.items[2], {GHR}, q);
      q += GHR;
      cbor_read_tensor(*fw.items[3], {GHR}, q);
      q += GHR;
      end = cbor_read_mlp(mod.at("second"), H, module->hidden, module->out_dim, q);
    }
    RL_REQUIRE((uint64_t)(end - p.data()) == module->P, "CBOR module: parameter count mismatch");
    h2d(module->eng, module->d_params, p.data(), module->P * sizeof(float));
  });
}

}  // extern "C"
