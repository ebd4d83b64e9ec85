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

// KindDef variant names in declaration order, element sizes in bytes (torch/serialize.rs:12-31; tch Kind::elt_size_in_bytes)
static const char *const kKindNames[16] = {"Uint8", "Int8", "Int16", "Int", "Int64", "Half", "Float", "Double",
                                           "ComplexHalf", "ComplexFloat", "ComplexDouble", "Bool", "QInt8", "QUInt8",
                                           "QInt32", "BFloat16"};
static const uint32_t kKindSizes[16] = {1, 1, 2, 4, 8, 2, 4, 8, 4, 8, 16, 1, 1, 1, 4, 2};

// TensorDef (torch/serialize.rs:62-81): the five fields in declaration order, `data` as a byte string
static void cbor_tensor_def(cbor::Writer &w, int kind, const int64_t *shape, size_t rank, bool requires_grad,
                            const void *data, size_t data_bytes) {
  w.map(5);
  w.key("kind");
  w.text(kKindNames[kind]);
  w.key("shape");
  w.array(rank);
  for (size_t i = 0; i < rank; ++i) w.sint(shape[i]);
  w.key("requires_grad");
  w.boolean(requires_grad);
  w.key("byte_order");
  w.text("LittleEndian");  // ByteOrder::native() of every machine this library runs on
  w.key("data");
  w.bytes(data, data_bytes);
}

static void cbor_tensor(cbor::Writer &w, const float *data, std::initializer_list<int64_t> shape) {
  size_t count = 1;
  for (int64_t d : shape) count *= (size_t)d;
  cbor_tensor_def(w, RL_KIND_FLOAT, shape.begin(), shape.size(), true, data, count * sizeof(float));
}

struct TensorDefView {
  int kind = -1;
  std::vector<int64_t> shape;
  bool requires_grad = false;
  const std::string *data = nullptr;
};

// the reader accepts what serde's derived Deserialize accepts: the five fields of the struct in ANY order (a map is
// keyed, not positional), each exactly once, known variant names, native byte order (`From<&TensorDef> for Tensor`
// asserts it, serialize.rs:110-114), consistent sizes
static TensorDefView cbor_parse_tensor_def(const cbor::Value &t) {
  static const char *const names[5] = {"kind", "shape", "requires_grad", "byte_order", "data"};
  RL_REQUIRE(t.kind == cbor::Value::MAP && t.fields.size() == 5, "CBOR tensor: expected the five TensorDef fields");
  for (int i = 0; i < 5; ++i) {
    int seen = 0;
    for (const auto &f : t.fields) seen += f.first == names[i];
    RL_REQUIRE(seen == 1, "CBOR tensor: every TensorDef field exactly once");
  }
  TensorDefView v;
  const cbor::Value &k = t.at("kind");
  RL_REQUIRE(k.kind == cbor::Value::TEXT, "CBOR tensor: kind must be a unit variant name");
  for (int i = 0; i < 16; ++i)
    if (k.s == kKindNames[i]) v.kind = i;
  RL_REQUIRE(v.kind >= 0, "CBOR tensor: unknown kind");
  const cbor::Value &bo = t.at("byte_order");
  RL_REQUIRE(bo.kind == cbor::Value::TEXT && (bo.s == "LittleEndian" || bo.s == "BigEndian"),
             "CBOR tensor: unknown byte order");
  RL_REQUIRE(bo.s == "LittleEndian", "CBOR tensor: data has non-native byte order");
  const cbor::Value &sh = t.at("shape");
  RL_REQUIRE(sh.kind == cbor::Value::ARRAY, "CBOR tensor: shape must be a sequence");
  uint64_t count = 1;
  for (auto &it : sh.items) {
    const int64_t d = it->as_int();
    RL_REQUIRE(d >= 0, "CBOR tensor: negative extent");
    v.shape.push_back(d);
    // the element count must stay a count: [2^32, 2^32] would wrap to 0 and pass the length check with empty data
    RL_REQUIRE(d == 0 || count <= (uint64_t(1) << 40) / (uint64_t)d, "CBOR tensor: the shape's element count overflows");
    count *= (uint64_t)d;
  }
  const cbor::Value &rg = t.at("requires_grad");
  RL_REQUIRE(rg.kind == cbor::Value::BOOL, "CBOR tensor: requires_grad must be a bool");
  v.requires_grad = rg.b;
  const cbor::Value &data = t.at("data");
  RL_REQUIRE(data.kind == cbor::Value::BYTES && data.s.size() == count * kKindSizes[v.kind],
             "CBOR tensor: bad data length");
  v.data = &data.s;
  return v;
}

// Mlp { layers, activation, output_activation } over `p` = [W1, b1, W2, b2] (ff/mlp.rs:45-50, ff/linear.rs:43-50)
// (`widths`: in, hidden sizes ..., out)
// unit variants of `Activation` as serde writes them: the variant's name (ff/activation.rs:11-20)
static const char *const kActivationNames[4] = {"Identity", "Relu", "Sigmoid", "Tanh"};
static void cbor_mlp_layers(cbor::Writer &w, const float *p, const std::vector<int64_t> &widths, int act = RL_ACT_RELU,
                            int out_act = RL_ACT_IDENTITY, bool has_bias = true) {
  w.map(3);
  w.key("layers");
  w.array(widths.size() - 1);
  for (size_t l = 0; l + 1 < widths.size(); ++l) {
    const int64_t in = widths[l], out = widths[l + 1];
    w.map(2);
    w.key("kernel");
    cbor_tensor(w, p, {out, in});
    p += in * out;
    w.key("bias");  // Option<TensorDef> (ff/linear.rs:45-50): None -> null
    if (has_bias) {
      cbor_tensor(w, p, {out});
      p += out;
    } else {
      w.null();
    }
  }
  w.key("activation");
  w.text(kActivationNames[act]);
  w.key("output_activation");
  w.text(kActivationNames[out_act]);
}
static void cbor_mlp(cbor::Writer &w, const float *p, int64_t in, int64_t hid, int64_t out) {
  cbor_mlp_layers(w, p, {in, hid, out});
}
static std::vector<int64_t> mlp_widths(const rl_mlp *m) {
  std::vector<int64_t> v{(int64_t)m->in_dim};
  for (uint32_t l = 0; l < m->n_hidden; ++l) v.push_back((int64_t)m->widths[l]);
  v.push_back((int64_t)m->out_dim);
  return v;
}

static void cbor_module(cbor::Writer &w, const rl_mlp *m, const std::vector<float> &p) {
  if (m->kind == RL_MODULE_MLP) {
    cbor_mlp_layers(w, p.data(), mlp_widths(m), m->act, m->out_act, m->has_bias);
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
  w.array((m->has_bias ? 4 : 2) * m->rnn_layers);  // per layer [w_ih, w_hh (, b_ih, b_hh)] (seq/rnn/mod.rs:223-257)
  const float *q = p.data();
  for (uint32_t l = 0; l < m->rnn_layers; ++l) {
    const int64_t K = l == 0 ? D : H;
    cbor_tensor(w, q, {GHR, K});
    q += GHR * K;
    cbor_tensor(w, q, {GHR, H});
    q += GHR * H;
    if (m->has_bias) {
      cbor_tensor(w, q, {GHR});
      q += GHR;
      cbor_tensor(w, q, {GHR});
      q += GHR;
    }
  }
  w.key("has_biases");
  w.boolean(m->has_bias);
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

// IndexedTypeSpace<T>: its only field is #[serde(skip)] (spaces/indexed_type.rs:57-64) -> a struct of length 0
static void cbor_indexed_type_space(cbor::Writer &w) { w.map(0); }

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
    cbor_indexed_type_space(w);
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
  const TensorDefView v = cbor_parse_tensor_def(t);
  RL_REQUIRE(v.kind == RL_KIND_FLOAT, "CBOR tensor: kind must be Float");
  RL_REQUIRE(v.shape.size() == shape.size(), "CBOR tensor: unexpected rank");
  size_t i = 0;
  for (int64_t d : shape) RL_REQUIRE(v.shape[i++] == d, "CBOR tensor: unexpected shape");
  std::memcpy(dst, v.data->data(), v.data->size());
}

static float *cbor_read_mlp_layers(const cbor::Value &m, const std::vector<int64_t> &widths, float *dst,
                                   int act = RL_ACT_RELU, int out_act = RL_ACT_IDENTITY, bool has_bias = true) {
  RL_REQUIRE(m.at("activation").s == kActivationNames[act] && m.at("output_activation").s == kActivationNames[out_act],
             "CBOR module: the document's activations are not the module's");
  const cbor::Value &layers = m.at("layers");
  RL_REQUIRE(layers.kind == cbor::Value::ARRAY && layers.items.size() + 1 == widths.size(),
             "CBOR module: the number of layers does not match the module");
  for (size_t l = 0; l + 1 < widths.size(); ++l) {
    const int64_t in = widths[l], out = widths[l + 1];
    const cbor::Value &lin = *layers.items[l];
    cbor_read_tensor(lin.at("kernel"), {out, in}, dst);
    dst += in * out;
    RL_REQUIRE((lin.at("bias").kind == cbor::Value::MAP) == has_bias,
               "CBOR module: the document's layers and the module disagree about bias vectors");
    if (has_bias) {
      cbor_read_tensor(lin.at("bias"), {out}, dst);
      dst += out;
    }
  }
  return dst;
}
static float *cbor_read_mlp(const cbor::Value &m, int64_t in, int64_t hid, int64_t out, float *dst) {
  return cbor_read_mlp_layers(m, {in, hid, out}, dst);
}

int32_t rl_module_from_cbor(rl_mlp *module, const uint8_t *buf, uint64_t len) {
  return guarded(module ? module->eng : nullptr, [&] {
    RL_REQUIRE(module && buf, "NULL argument");
    cbor::ValuePtr doc = cbor::Reader(buf, (size_t)len).parse();
    const cbor::Value &mod = doc->has("policy_module") ? doc->at("policy_module") : doc->at("action_value_fn");
    std::vector<float> p(module->P);
    float *end;
    if (module->kind == RL_MODULE_MLP) {
      end = cbor_read_mlp_layers(mod, mlp_widths(module), p.data(), module->act, module->out_act, module->has_bias);
    } else {
      const int64_t H = module->gru_hidden, D = module->in_dim, GHR = (int64_t)rl_module_gates(module->kind) * H;
      RL_REQUIRE(mod.at("activation").s == "Relu", "CBOR module: Chain activation must be Relu");
      const cbor::Value &rnn = mod.at("first");
      RL_REQUIRE(rnn.at("hidden_size").as_int() == H, "CBOR module: GRU hidden size mismatch");
      RL_REQUIRE(rnn.at("dropout").as_float() == 0.0, "CBOR module: dropout is not built");
      const cbor::Value &wts = rnn.at("weights");
      RL_REQUIRE(wts.at("has_biases").kind == cbor::Value::BOOL && wts.at("has_biases").b == module->has_bias,
                 "CBOR module: has_biases does not match the module");
      const cbor::Value &fw = wts.at("flat_weights");
      const size_t per = module->has_bias ? 4 : 2;
      RL_REQUIRE(fw.kind == cbor::Value::ARRAY && fw.items.size() == per * module->rnn_layers,
                 "CBOR module: the recurrent layer count does not match the module");
      float *q = p.data();
      for (uint32_t l = 0; l < module->rnn_layers; ++l) {
        const int64_t K = l == 0 ? D : H;
        cbor_read_tensor(*fw.items[per * l + 0], {GHR, K}, q);
        q += GHR * K;
        cbor_read_tensor(*fw.items[per * l + 1], {GHR, H}, q);
        q += GHR * H;
        if (module->has_bias) {
          cbor_read_tensor(*fw.items[per * l + 2], {GHR}, q);
          q += GHR;
          cbor_read_tensor(*fw.items[per * l + 3], {GHR}, q);
          q += GHR;
        }
      }
      end = cbor_read_mlp(mod.at("second"), H, module->hidden, module->out_dim, q);
    }
    RL_REQUIRE((uint64_t)(end - p.data()) == module->P, "CBOR module: parameter count mismatch");
    h2d(module->eng, module->d_params, p.data(), module->P * sizeof(float));
    wimg_invalidate(module);  // (the parameters changed under the module's weight image, bf16_tile.hpp)
  });
}

int32_t rl_tensor_def_to_cbor(int32_t kind, const int64_t *shape, uint32_t rank, int32_t requires_grad,
                              const void *data, uint64_t data_bytes, uint8_t *buf, uint64_t cap, uint64_t *len_out) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(len_out, "len_out is NULL");
    RL_REQUIRE(kind >= 0 && kind < 16, "unknown tensor kind");
    RL_REQUIRE(rank == 0 || shape, "shape is NULL");
    uint64_t count = 1;
    for (uint32_t i = 0; i < rank; ++i) {
      RL_REQUIRE(shape[i] >= 0, "negative extent");
      count *= (uint64_t)shape[i];
    }
    RL_REQUIRE(data_bytes == count * kKindSizes[kind], "data length does not match shape x element size");
    RL_REQUIRE(data_bytes == 0 || data, "data is NULL");
    cbor::Writer w;
    cbor_tensor_def(w, kind, shape, rank, requires_grad != 0, data, (size_t)data_bytes);
    *len_out = w.out.size();
    if (buf != nullptr) {
      RL_REQUIRE(cap >= w.out.size(), "buffer too small for the CBOR document");
      std::memcpy(buf, w.out.data(), w.out.size());
    }
  });
}

int32_t rl_tensor_def_from_cbor(const uint8_t *buf, uint64_t len, int32_t *kind_out, int64_t *shape_out,
                                uint32_t shape_cap, uint32_t *rank_out, int32_t *requires_grad_out, void *data_out,
                                uint64_t data_cap, uint64_t *data_bytes_out) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(buf, "buf is NULL");
    cbor::ValuePtr doc = cbor::Reader(buf, (size_t)len).parse();
    const TensorDefView v = cbor_parse_tensor_def(*doc);
    if (kind_out) *kind_out = v.kind;
    if (rank_out) *rank_out = (uint32_t)v.shape.size();
    if (requires_grad_out) *requires_grad_out = v.requires_grad ? 1 : 0;
    if (data_bytes_out) *data_bytes_out = v.data->size();
    if (shape_out) {
      RL_REQUIRE(shape_cap >= v.shape.size(), "shape buffer too small");
      for (size_t i = 0; i < v.shape.size(); ++i) shape_out[i] = v.shape[i];
    }
    if (data_out) {
      RL_REQUIRE(data_cap >= v.data->size(), "data buffer too small");
      std::memcpy(data_out, v.data->data(), v.data->size());
    }
  });
}

int32_t rl_indexed_type_space_to_cbor(uint8_t *buf, uint64_t cap, uint64_t *len_out) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(len_out, "len_out is NULL");
    cbor::Writer w;
    cbor_indexed_type_space(w);
    *len_out = w.out.size();
    if (buf != nullptr) {
      RL_REQUIRE(cap >= w.out.size(), "buffer too small for the CBOR document");
      std::memcpy(buf, w.out.data(), w.out.size());
    }
  });
}

}  // extern "C"
