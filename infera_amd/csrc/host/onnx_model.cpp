#include "onnx_model.hpp"

#include <cstring>
#include <fstream>
#include <functional>

#include "common.hpp"

namespace infera_hip::onnx {

namespace {

// Cursor over a protobuf-encoded byte range.  Every read is bounds-checked; a violation throws.
class Wire {
 public:
  Wire(const uint8_t *b, const uint8_t *e) : p_(b), end_(e) {}
  bool done() const { return p_ >= end_; }

  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      need(1);
      uint8_t c = *p_++;
      v |= uint64_t(c & 0x7F) << shift;
      if (!(c & 0x80)) return v;
    }
    fail("varint too long");
  }
  // Returns field number, sets wire type.
  uint32_t key(int &wt) {
    uint64_t k = varint();
    wt = int(k & 7);
    if ((k >> 3) == 0) fail("field number 0");
    return uint32_t(k >> 3);
  }
  Wire sub() {
    uint64_t n = varint();
    if (n > uint64_t(end_ - p_)) fail("length-delimited field overruns buffer");
    Wire w(p_, p_ + n);
    p_ += n;
    return w;
  }
  std::string str() {
    Wire w = sub();
    return std::string(reinterpret_cast<const char *>(w.p_), size_t(w.end_ - w.p_));
  }
  uint32_t fixed32() {
    need(4);
    uint32_t v;
    std::memcpy(&v, p_, 4);
    p_ += 4;
    return v;
  }
  void skip(int wt) {
    switch (wt) {
      case 0: (void)varint(); break;
      case 1: need(8); p_ += 8; break;
      case 2: (void)sub(); break;
      case 5: need(4); p_ += 4; break;
      default: fail("unsupported wire type");
    }
  }
  const uint8_t *data() const { return p_; }
  size_t size() const { return size_t(end_ - p_); }

  // Visits every field: fn(field, wire_type, cursor) returns true if it consumed the value.
  void each(const std::function<bool(uint32_t, int, Wire &)> &fn) {
    while (!done()) {
      int wt;
      uint32_t f = key(wt);
      if (!fn(f, wt, *this)) skip(wt);
    }
  }

 private:
  [[noreturn]] static void fail(const char *what) { throw InferaError::onnx(std::string("protobuf decode: ") + what); }
  void need(size_t n) const {
    if (size_t(end_ - p_) < n) fail("truncated input");
  }
  const uint8_t *p_, *end_;
};

void read_packed_or_single_i64(Wire &w, int wt, std::vector<int64_t> &out) {
  if (wt == 2) {
    Wire s = w.sub();
    while (!s.done()) out.push_back(int64_t(s.varint()));
  } else {
    out.push_back(int64_t(w.varint()));
  }
}

void read_packed_or_single_f32(Wire &w, int wt, std::vector<float> &out) {
  if (wt == 2) {
    Wire s = w.sub();
    size_t n = s.size() / 4;
    size_t base = out.size();
    out.resize(base + n);
    std::memcpy(out.data() + base, s.data(), n * 4);
  } else if (wt == 5) {
    uint32_t u = w.fixed32();
    float f;
    std::memcpy(&f, &u, 4);
    out.push_back(f);
  } else {
    w.skip(wt);
  }
}

// TensorProto: dims=1 data_type=2 float_data=4 int32_data=5 int64_data=7 name=8 raw_data=9 double_data=10
std::shared_ptr<TensorData> read_tensor(Wire w) {
  auto t = std::make_shared<TensorData>();
  int dtype = 0;
  std::vector<float> fdata;
  std::vector<int64_t> idata;
  std::vector<double> ddata;
  const uint8_t *raw = nullptr;
  size_t rawlen = 0;
  bool external = false;
  w.each([&](uint32_t f, int wt, Wire &c) {
    switch (f) {
      case 1: read_packed_or_single_i64(c, wt, t->dims); return true;
      case 2: if (wt != 0) return false; dtype = int(c.varint()); return true;
      case 4: read_packed_or_single_f32(c, wt, fdata); return true;
      case 5:
      case 7: read_packed_or_single_i64(c, wt, idata); return true;
      case 8: if (wt != 2) return false; t->name = c.str(); return true;
      case 9: {
        if (wt != 2) return false;
        Wire s = c.sub();
        raw = s.data();
        rawlen = s.size();
        return true;
      }
      case 10: {
        if (wt != 2) return false;
        Wire s = c.sub();
        size_t n = s.size() / 8;
        ddata.resize(n);
        std::memcpy(ddata.data(), s.data(), n * 8);
        return true;
      }
      case 14: if (wt != 0) return false; external = c.varint() == 1; return true;
      default: return false;
    }
  });
  if (external) throw InferaError::onnx("tensor '" + t->name + "' uses external data, which is not supported");
  for (auto d : t->dims)
    if (d < 0) throw InferaError::onnx("tensor '" + t->name + "' has a negative dimension");
  auto size_err = [&] { return InferaError::onnx("tensor '" + t->name + "': element count does not match dims"); };
  // Element count with overflow detection: dims such as [4, 2^62] wrap a plain product to 0 and would
  // pass every size comparison below.  The count is then compared with the payload that is really in the
  // file BEFORE anything is allocated, so no declared shape can make the loader reserve more than it read.
  size_t n = 1;
  for (auto d : t->dims)
    if (__builtin_mul_overflow(n, size_t(d), &n)) throw size_err();
  if (n == 0)  // an empty tensor must not smuggle an absurd extent past the payload check either
    for (auto d : t->dims)
      if (d > (int64_t(1) << 31)) throw size_err();
  switch (dtype) {
    case kFloat:
      t->dtype = kFloat;
      if (raw) {
        if (rawlen / 4 != n || rawlen % 4) throw size_err();
        t->f32.resize(n);
        std::memcpy(t->f32.data(), raw, rawlen);
      } else {
        if (fdata.size() != n) throw size_err();
        t->f32 = std::move(fdata);
      }
      break;
    case kDouble:
      t->dtype = kFloat;  // narrowed: the whole path computes in f32
      if (raw) {
        if (rawlen / 8 != n || rawlen % 8) throw size_err();
        t->f32.resize(n);
        for (size_t i = 0; i < n; i++) {
          double d;
          std::memcpy(&d, raw + i * 8, 8);
          t->f32[i] = float(d);
        }
      } else {
        if (ddata.size() != n) throw size_err();
        t->f32.resize(n);
        for (size_t i = 0; i < n; i++) t->f32[i] = float(ddata[i]);
      }
      break;
    case kInt64:
    case kInt32:
      t->dtype = kInt64;
      if (raw) {
        const size_t es = dtype == kInt64 ? 8 : 4;
        if (rawlen / es != n || rawlen % es) throw size_err();
        t->i64.resize(n);
        for (size_t i = 0; i < n; i++) {
          if (es == 8) {
            int64_t v;
            std::memcpy(&v, raw + i * 8, 8);
            t->i64[i] = v;
          } else {
            int32_t v;
            std::memcpy(&v, raw + i * 4, 4);
            t->i64[i] = v;
          }
        }
      } else {
        if (idata.size() != n) throw size_err();
        t->i64 = std::move(idata);
      }
      break;
    default:
      throw InferaError::onnx("tensor '" + t->name + "': unsupported data_type " + std::to_string(dtype));
  }
  return t;
}

// AttributeProto: name=1 f=2 i=3 s=4 t=5 floats=7 ints=8 type=20
Attribute read_attribute(Wire w) {
  Attribute a;
  bool saw_f = false, saw_i = false;
  w.each([&](uint32_t f, int wt, Wire &c) {
    switch (f) {
      case 1: if (wt != 2) return false; a.name = c.str(); return true;
      case 2: {
        if (wt != 5) return false;
        uint32_t u = c.fixed32();
        std::memcpy(&a.f, &u, 4);
        saw_f = true;
        return true;
      }
      case 3: if (wt != 0) return false; a.i = int64_t(c.varint()); saw_i = true; return true;
      case 4: if (wt != 2) return false; a.s = c.str(); return true;
      case 5: if (wt != 2) return false; a.t = read_tensor(c.sub()); return true;
      case 7: read_packed_or_single_f32(c, wt, a.floats); return true;
      case 8: read_packed_or_single_i64(c, wt, a.ints); return true;
      case 20: if (wt != 0) return false; a.type = int(c.varint()); return true;
      default: return false;
    }
  });
  if (a.type == 0) {  // writers older than IR 3 omit `type`
    if (!a.ints.empty()) a.type = 7;
    else if (!a.floats.empty()) a.type = 6;
    else if (a.t) a.type = 4;
    else if (!a.s.empty()) a.type = 3;
    else if (saw_f) a.type = 1;
    else if (saw_i) a.type = 2;
  }
  return a;
}

// NodeProto: input=1 output=2 name=3 op_type=4 attribute=5 domain=7
NodeDef read_node(Wire w) {
  NodeDef n;
  w.each([&](uint32_t f, int wt, Wire &c) {
    if (wt != 2) return false;
    switch (f) {
      case 1: n.inputs.push_back(c.str()); return true;
      case 2: n.outputs.push_back(c.str()); return true;
      case 3: n.name = c.str(); return true;
      case 4: n.op = c.str(); return true;
      case 5: {
        Attribute a = read_attribute(c.sub());
        n.attrs[a.name] = std::move(a);
        return true;
      }
      case 7: n.domain = c.str(); return true;
      default: return false;
    }
  });
  if (n.op.empty()) throw InferaError::onnx("node without op_type");
  return n;
}

// ValueInfoProto: name=1 type=2{tensor_type=1{elem_type=1 shape=2{dim=1{dim_value=1 dim_param=2}}}}
ValueDef read_value_info(Wire w) {
  ValueDef v;
  w.each([&](uint32_t f, int wt, Wire &c) {
    if (f == 1 && wt == 2) {
      v.name = c.str();
      return true;
    }
    if (f == 2 && wt == 2) {
      c.sub().each([&](uint32_t f2, int wt2, Wire &c2) {
        if (f2 != 1 || wt2 != 2) return false;
        c2.sub().each([&](uint32_t f3, int wt3, Wire &c3) {
          if (f3 == 1 && wt3 == 0) {
            v.elem_type = int(c3.varint());
            return true;
          }
          if (f3 == 2 && wt3 == 2) {
            v.has_shape = true;
            c3.sub().each([&](uint32_t f4, int wt4, Wire &c4) {
              if (f4 != 1 || wt4 != 2) return false;
              int64_t dim = -1;
              c4.sub().each([&](uint32_t f5, int wt5, Wire &c5) {
                if (f5 == 1 && wt5 == 0) {
                  dim = int64_t(c5.varint());
                  return true;
                }
                return false;
              });
              v.dims.push_back(dim);
              return true;
            });
            return true;
          }
          return false;
        });
        return true;
      });
      return true;
    }
    return false;
  });
  return v;
}

// GraphProto: node=1 name=2 initializer=5 input=11 output=12
void read_graph(Wire w, Model &m) {
  std::vector<ValueDef> declared_inputs;
  w.each([&](uint32_t f, int wt, Wire &c) {
    if (wt != 2) return false;
    switch (f) {
      case 1: m.nodes.push_back(read_node(c.sub())); return true;
      case 2: m.graph_name = c.str(); return true;
      case 5: {
        auto t = read_tensor(c.sub());
        m.initializers[t->name] = t;
        return true;
      }
      case 11: declared_inputs.push_back(read_value_info(c.sub())); return true;
      case 12: m.outputs.push_back(read_value_info(c.sub())); return true;
      default: return false;
    }
  });
  for (auto &v : declared_inputs)
    if (!m.initializers.count(v.name)) m.inputs.push_back(std::move(v));
}

}  // namespace

Model parse_bytes(const uint8_t *data, size_t len) {
  Model m;
  bool saw_graph = false;
  Wire w(data, data + len);
  // ModelProto: ir_version=1 producer_name=2 graph=7 opset_import=8{domain=1 version=2}
  w.each([&](uint32_t f, int wt, Wire &c) {
    if (f == 1 && wt == 0) {
      m.ir_version = int64_t(c.varint());
      return true;
    }
    if (f == 2 && wt == 2) {
      m.producer = c.str();
      return true;
    }
    if (f == 7 && wt == 2) {
      read_graph(c.sub(), m);
      saw_graph = true;
      return true;
    }
    if (f == 8 && wt == 2) {
      std::string domain;
      int64_t version = 0;
      c.sub().each([&](uint32_t f2, int wt2, Wire &c2) {
        if (f2 == 1 && wt2 == 2) {
          domain = c2.str();
          return true;
        }
        if (f2 == 2 && wt2 == 0) {
          version = int64_t(c2.varint());
          return true;
        }
        return false;
      });
      if (domain.empty() || domain == "ai.onnx") m.opset = version;
      return true;
    }
    return false;
  });
  if (!saw_graph) throw InferaError::onnx("file is not an ONNX ModelProto (no graph)");
  if (m.inputs.empty()) throw InferaError::onnx("model has no runtime input");
  if (m.outputs.empty()) throw InferaError::onnx("model has no output");
  return m;
}

Model parse_file(const std::string &path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw InferaError::onnx("cannot open model file '" + path + "'");
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (buf.empty()) throw InferaError::onnx("model file '" + path + "' is empty");
  return parse_bytes(buf.data(), buf.size());
}

}  // namespace infera_hip::onnx
