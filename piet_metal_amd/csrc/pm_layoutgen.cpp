// pm_layoutgen: layout code generator with a HIP / C++ target.
//
// The reference keeps host and device layouts in step with a proc-macro,
// piet-gpu-derive (piet-gpu-derive/src/lib.rs), fed by the `piet_gpu!` module descriptions of
// src/main.rs:11-93; it has MSL and HLSL targets (lib.rs:24-27), and the enum loaders and the
// writers are left as TODOs in comments (lib.rs:1051-1117).  This tool reads the same grammar
//
//     mod NAME { struct S { field: TYPE, ... }  enum E { Variant(S), Unit, Variant(S) = 7, ... } }
//     TYPE = u8 | u16 | u32 | i8 | i16 | i32 | f32 | [SCALAR; N] | Ref<T>
//
// and prints ONE header usable from hipcc (device + host) and g++ (host): per struct a packed
// mirror with static_asserted offsets, `S_read`, per-field getters, `S_SIZE`; per enum its tags,
// `E_SIZE`, `E_tag`, the generic record, and -- the parts the reference never finished --
// `Variant_load(const E &)`, `S_write(buf, ref, s)` and `E_write_tag`.
//
// Layout rules (lib.rs:1-5, and what TestApp/GenTypes.h shows): a struct that appears as an enum
// variant starts with the u32 tag; scalars are naturally aligned; a vector [T; N] is aligned to
// its size (<= 16); a struct is as aligned as its widest member (>= 4) and padded to that; an
// enum is as large as its largest variant.
//
//     pm_layoutgen piet_layout.pgpu > pm_layout_gen.h
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Type {
    enum Kind { kScalar, kVector, kRef } kind = kScalar;
    std::string scalar;  // u8 .. f32 (element type for vectors)
    int n = 1;
    std::string target;  // Ref<target>
};

struct Field {
    std::string name;
    Type type;
    size_t offset = 0;
};

struct Struct {
    std::string name;
    std::vector<Field> fields;
    bool is_variant = false;
    size_t size = 0, align = 4;
};

struct Variant {
    std::string name, payload;  // payload struct name, "" for a unit variant
    unsigned tag = 0;
};

struct Enum {
    std::string name;
    std::vector<Variant> variants;
    size_t size = 4;
};

struct Module {
    std::string name;
    std::vector<std::string> order;  // definition order, names
    std::map<std::string, Struct> structs;
    std::map<std::string, Enum> enums;
};

[[noreturn]] void Die(const std::string &msg) {
    std::fprintf(stderr, "pm_layoutgen: %s\n", msg.c_str());
    std::exit(2);
}

// ---- lexer ---------------------------------------------------------------------------------
struct Lexer {
    std::vector<std::string> tok;
    size_t pos = 0;
    explicit Lexer(const std::string &src) {
        for (size_t i = 0; i < src.size();) {
            const char c = src[i];
            if (std::isspace(static_cast<unsigned char>(c))) {
                ++i;
            } else if (c == '/' && i + 1 < src.size() && src[i + 1] == '/') {
                while (i < src.size() && src[i] != '\n') ++i;
            } else if (std::isalnum(static_cast<unsigned char>(c)) || c == '_') {
                size_t j = i;
                while (j < src.size() && (std::isalnum(static_cast<unsigned char>(src[j])) || src[j] == '_')) ++j;
                tok.push_back(src.substr(i, j - i));
                i = j;
            } else {
                tok.push_back(std::string(1, c));
                ++i;
            }
        }
    }
    bool done() const { return pos >= tok.size(); }
    const std::string &peek() const {
        static const std::string eof = "<eof>";
        return done() ? eof : tok[pos];
    }
    std::string next() {
        if (done()) Die("unexpected end of input");
        return tok[pos++];
    }
    void expect(const std::string &t) {
        const std::string got = next();
        if (got != t) Die("expected '" + t + "', got '" + got + "'");
    }
    bool accept(const std::string &t) {
        if (!done() && tok[pos] == t) {
            ++pos;
            return true;
        }
        return false;
    }
};

bool IsScalar(const std::string &s) {
    static const std::set<std::string> k = {"u8", "u16", "u32", "i8", "i16", "i32", "f32"};
    return k.count(s) != 0;
}
size_t ScalarSize(const std::string &s) { return s == "u8" || s == "i8" ? 1 : (s == "u16" || s == "i16" ? 2 : 4); }
std::string ScalarC(const std::string &s) {
    if (s == "f32") return "float";
    return std::string(s[0] == 'u' ? "uint" : "int") + s.substr(1) + "_t";
}

Type ParseType(Lexer &lx) {
    Type t;
    if (lx.accept("[")) {
        t.kind = Type::kVector;
        t.scalar = lx.next();
        if (!IsScalar(t.scalar)) Die("vector of non-scalar " + t.scalar);
        lx.expect(";");
        t.n = std::atoi(lx.next().c_str());
        if (t.n < 1 || t.n > 4) Die("vector length must be 1..4");
        lx.expect("]");
    } else {
        const std::string id = lx.next();
        if (id == "Ref") {
            t.kind = Type::kRef;
            lx.expect("<");
            t.target = lx.next();
            lx.expect(">");
        } else if (IsScalar(id)) {
            t.scalar = id;
        } else {
            Die("unknown type " + id);
        }
    }
    return t;
}

size_t TypeSize(const Type &t) { return t.kind == Type::kRef ? 4 : ScalarSize(t.scalar) * static_cast<size_t>(t.n); }
size_t TypeAlign(const Type &t) { return std::min<size_t>(TypeSize(t), 16); }
std::string TypeC(const Type &t) {
    if (t.kind == Type::kRef) return t.target + "Ref";
    if (t.kind == Type::kVector) return "pm_" + t.scalar + "x" + std::to_string(t.n);
    return ScalarC(t.scalar);
}

std::vector<Module> Parse(const std::string &src) {
    Lexer lx(src);
    std::vector<Module> mods;
    while (!lx.done()) {
        lx.accept("piet_gpu");  // the description may keep the macro wrapper: piet_gpu! { mod ... }
        lx.accept("!");
        const bool wrapped = lx.accept("{");
        lx.expect("mod");
        Module m;
        m.name = lx.next();
        lx.expect("{");
        while (!lx.accept("}")) {
            const std::string kw = lx.next();
            if (kw == "struct") {
                Struct s;
                s.name = lx.next();
                lx.expect("{");
                while (!lx.accept("}")) {
                    Field f;
                    f.name = lx.next();
                    lx.expect(":");
                    f.type = ParseType(lx);
                    lx.accept(",");
                    s.fields.push_back(f);
                }
                m.order.push_back(s.name);
                m.structs[s.name] = s;
            } else if (kw == "enum") {
                Enum e;
                e.name = lx.next();
                lx.expect("{");
                unsigned next_tag = 1;
                while (!lx.accept("}")) {
                    Variant v;
                    v.name = lx.next();
                    if (lx.accept("(")) {
                        v.payload = lx.next();
                        lx.expect(")");
                    }
                    if (lx.accept("=")) next_tag = static_cast<unsigned>(std::atoi(lx.next().c_str()));
                    v.tag = next_tag++;
                    lx.accept(",");
                    e.variants.push_back(v);
                }
                m.order.push_back(e.name);
                m.enums[e.name] = e;
            } else {
                Die("expected struct or enum, got " + kw);
            }
        }
        if (wrapped) lx.expect("}");
        mods.push_back(m);
    }
    return mods;
}

size_t AlignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

void Layout(Module &m) {
    for (auto &e : m.enums)
        for (const Variant &v : e.second.variants)
            if (!v.payload.empty()) {
                if (!m.structs.count(v.payload)) Die("variant payload " + v.payload + " is not a struct of mod " + m.name);
                m.structs[v.payload].is_variant = true;
            }
    for (auto &kv : m.structs) {
        Struct &s = kv.second;
        size_t off = s.is_variant ? 4 : 0;
        s.align = 4;
        for (Field &f : s.fields) {
            if (f.type.kind == Type::kRef && !m.structs.count(f.type.target) && !m.enums.count(f.type.target) && !IsScalar(f.type.target))
                Die("Ref<" + f.type.target + "> names nothing in mod " + m.name);
            off = AlignUp(off, TypeAlign(f.type));
            f.offset = off;
            off += TypeSize(f.type);
            s.align = std::max(s.align, TypeAlign(f.type));
        }
        s.size = AlignUp(std::max<size_t>(off, 4), 4);  // records are arrays of dwords
    }
    for (auto &kv : m.enums) {
        Enum &e = kv.second;
        e.size = 4;
        for (const Variant &v : e.variants)
            if (!v.payload.empty()) e.size = std::max(e.size, m.structs[v.payload].size);
        e.size = AlignUp(e.size, 8);  // items and commands are read as 64-bit words
    }
}

std::string Upper(const std::string &camel) {  // PietItem -> PIET_ITEM (to_snake_case().to_uppercase(), lib.rs:1262)
    std::string r;
    for (size_t i = 0; i < camel.size(); ++i) {
        if (i && std::isupper(static_cast<unsigned char>(camel[i])) && !std::isupper(static_cast<unsigned char>(camel[i - 1]))) r += '_';
        r += static_cast<char>(std::toupper(static_cast<unsigned char>(camel[i])));
    }
    return r;
}

void Emit(const std::vector<Module> &mods, const std::string &source_name, std::ostream &o) {
    o << "// GENERATED by pm_layoutgen from " << source_name << " -- do not edit.\n"
      << "// (the HIP / C++ target of the reference's layout generator: piet-gpu-derive/src/lib.rs,\n"
      << "//  fed by the `piet_gpu!` descriptions of src/main.rs:11-93; tests/test_layoutgen_cpu.py checks that\n"
      << "//  this committed file is what the tool prints today)\n"
      << "#pragma once\n\n#include <cstddef>\n#include <cstdint>\n#include <cstring>\n\n"
      << "#if defined(__HIPCC__)\n#define PM_GEN_FN __host__ __device__ inline\n#else\n#define PM_GEN_FN inline\n#endif\n\n"
      << "namespace pm {\nnamespace gen {\n\n";
    std::set<std::string> vecs;
    for (const Module &m : mods)
        for (const auto &kv : m.structs)
            for (const Field &f : kv.second.fields)
                if (f.type.kind == Type::kVector) vecs.insert(f.type.scalar + " " + std::to_string(f.type.n));
    for (const std::string &v : vecs) {
        std::istringstream is(v);
        std::string sc;
        int n;
        is >> sc >> n;
        o << "struct pm_" << sc << "x" << n << " {\n    " << ScalarC(sc) << " v[" << n << "];\n};\n";
    }
    o << "\ntemplate <typename T>\nPM_GEN_FN T pm_gen_get(const uint8_t *p) {\n    T v;\n    memcpy(&v, p, sizeof(T));\n    return v;\n}\n"
      << "template <typename T>\nPM_GEN_FN void pm_gen_put(uint8_t *p, const T &v) {\n    memcpy(p, &v, sizeof(T));\n}\n\n";
    for (const Module &m : mods) {
        o << "// ---- mod " << m.name << " " << std::string(70 - m.name.size(), '-') << "\nnamespace " << m.name << " {\n\n";
        for (const std::string &n : m.order) o << "typedef uint32_t " << n << "Ref;\n";
        if (m.structs.count("f32") == 0) o << "typedef uint32_t f32Ref;\n";
        o << "\n";
        for (const std::string &n : m.order) {
            if (m.structs.count(n)) {
                const Struct &s = m.structs.at(n);
                o << "struct " << n << "Packed {\n";
                if (s.is_variant) o << "    uint32_t tag;\n";
                size_t off = s.is_variant ? 4 : 0;
                int pad = 0;
                for (const Field &f : s.fields) {
                    if (f.offset > off) o << "    uint8_t pad" << pad++ << "_[" << (f.offset - off) << "];\n";
                    o << "    " << TypeC(f.type) << " " << f.name << ";\n";
                    off = f.offset + TypeSize(f.type);
                }
                if (s.size > off) o << "    uint8_t pad" << pad++ << "_[" << (s.size - off) << "];\n";
                if (!s.is_variant && s.fields.empty()) o << "    uint32_t empty_;\n";
                o << "};\n";
                o << "static_assert(sizeof(" << n << "Packed) == " << s.size << ", \"" << n << " is " << s.size << " bytes\");\n";
                for (const Field &f : s.fields)
                    o << "static_assert(offsetof(" << n << "Packed, " << f.name << ") == " << f.offset << ", \"" << n << "." << f.name << "\");\n";
                o << "constexpr uint32_t " << Upper(n) << "_SIZE = " << s.size << ";\n";
                o << "PM_GEN_FN " << n << "Packed " << n << "_read(const uint8_t *buf, " << n << "Ref ref) { return pm_gen_get<" << n
                  << "Packed>(buf + ref); }\n";
                o << "PM_GEN_FN void " << n << "_write(uint8_t *buf, " << n << "Ref ref, const " << n << "Packed &s) { pm_gen_put(buf + ref, s); }\n";
                for (const Field &f : s.fields) {
                    o << "constexpr uint32_t " << n << "_" << f.name << "_OFFSET = " << f.offset << ";\n";
                    o << "PM_GEN_FN " << TypeC(f.type) << " " << n << "_" << f.name << "(const uint8_t *buf, " << n << "Ref ref) { return pm_gen_get<"
                      << TypeC(f.type) << ">(buf + ref + " << f.offset << "); }\n";
                }
                o << "\n";
            } else {
                const Enum &e = m.enums.at(n);
                o << "struct " << n << " {\n    uint32_t tag;\n    uint32_t body[" << (e.size - 4) / 4 << "];\n};\n";
                o << "static_assert(sizeof(" << n << ") == " << e.size << ", \"" << n << " is " << e.size << " bytes\");\n";
                o << "constexpr uint32_t " << Upper(n) << "_SIZE = " << e.size << ";\n";
                o << "PM_GEN_FN " << n << " " << n << "_read(const uint8_t *buf, " << n << "Ref ref) { return pm_gen_get<" << n << ">(buf + ref); }\n";
                o << "PM_GEN_FN uint32_t " << n << "_tag(const uint8_t *buf, " << n << "Ref ref) { return pm_gen_get<uint32_t>(buf + ref); }\n";
                o << "PM_GEN_FN void " << n << "_write_tag(uint8_t *buf, " << n << "Ref ref, uint32_t tag) { pm_gen_put(buf + ref, tag); }\n";
                for (const Variant &v : e.variants) {
                    o << "constexpr uint32_t " << n << "_" << v.name << " = " << v.tag << ";\n";
                    if (v.payload.empty()) continue;
                    const Struct &s = m.structs.at(v.payload);
                    // the loader the reference left in a comment (lib.rs:1051-1081): variant view of a generic record
                    o << "PM_GEN_FN " << v.payload << "Packed " << v.payload << "_load(const " << n << " &s) {\n    " << v.payload
                      << "Packed r;\n    memcpy(&r, &s, sizeof(r));\n    return r;\n}\n";
                    // and the packer: fields -> generic record, tag included, padding zeroed
                    o << "PM_GEN_FN " << n << " " << n << "_" << v.name << "_pack(";
                    bool first = true;
                    for (const Field &f : s.fields) {
                        o << (first ? "" : ", ") << TypeC(f.type) << " " << f.name;
                        first = false;
                    }
                    o << ") {\n    " << n << " c;\n    memset(&c, 0, sizeof(c));\n    c.tag = " << n << "_" << v.name << ";\n";
                    for (const Field &f : s.fields)
                        o << "    memcpy(reinterpret_cast<uint8_t *>(&c) + " << f.offset << ", &" << f.name << ", " << TypeSize(f.type) << ");\n";
                    o << "    return c;\n}\n";
                }
                o << "\n";
            }
        }
        o << "}  // namespace " << m.name << "\n\n";
    }
    o << "}  // namespace gen\n}  // namespace pm\n";
}

}  // namespace

int main(int argc, char **argv) {
    if (argc != 2) {
        std::fprintf(stderr, "usage: pm_layoutgen <description.pgpu>   (header on stdout)\n");
        return 2;
    }
    std::ifstream in(argv[1]);
    if (!in) Die(std::string("cannot read ") + argv[1]);
    std::stringstream ss;
    ss << in.rdbuf();
    std::vector<Module> mods = Parse(ss.str());
    for (Module &m : mods) Layout(m);
    std::string name = argv[1];
    const size_t slash = name.find_last_of('/');
    if (slash != std::string::npos) name = name.substr(slash + 1);
    Emit(mods, "piet_metal_amd/layout/" + name, std::cout);
    return 0;
}
