// Context-free grammars for constrained decoding (vLLM's `guided_grammar`).
//
// Input: EBNF in the GBNF spelling -
//     root  ::= expr
//     expr  ::= term (("+" | "-") term)*
//     term  ::= [0-9]+ | "(" expr ")"
// with "literals" (escapes \n \t \r \\ \" \xHH), [character classes] (ranges, ^ negation, the same
// escapes), rule references, ( groups ), | alternation, the postfix operators * + ? and {m}, {m,},
// {m,n}, `.` for any byte, and # comments.  The start rule is `root`.
//
// The grammar works on BYTES: a literal with non-ASCII characters is its UTF-8 bytes, a negated class
// accepts every byte outside it.  Repetitions and groups are rewritten into helper rules, so a compiled
// grammar is a list of rules, each a list of alternatives, each a sequence of symbols that are either a
// 256-bit byte set or a rule reference.
//
// Matching keeps a SET OF STACKS (the text so far may be derivable in several ways).  A stack lists the
// positions still to be matched, innermost last, and is always kept "expanded": its innermost position
// points at a byte set (or the stack is empty - the text is a complete sentence).  Accepting a byte
// advances every stack whose byte set contains it and re-expands the results; rule references fork one
// stack per alternative.  Left recursion would expand forever and is refused when the grammar is
// compiled.
//
// The token mask walks a trie of the vocabulary's byte strings with the stack set, so tokens that share
// a prefix share the work and a byte no stack accepts prunes its whole subtree.  Stack sets met on
// the way are interned and their byte transitions remembered on the grammar (a DFA built lazily over
// the sets actually reached): inside a string literal, say, every trie node after the first costs one
// hash lookup.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

using ByteSet = std::array<uint64_t, 4>;
inline bool bs_has(const ByteSet& s, unsigned c) { return (s[c >> 6] >> (c & 63)) & 1; }
inline void bs_add(ByteSet& s, unsigned c) { s[c >> 6] |= 1ull << (c & 63); }

struct Sym {
  int rule;   // >= 0: reference to that rule; -1: byte set `set`
  int set;
};
using Alt = std::vector<Sym>;
struct Rule {
  std::string name;
  std::vector<Alt> alts;
};

struct Grammar {
  std::vector<Rule> rules;
  std::vector<ByteSet> sets;
  int root = -1;
};

struct Pos {
  int rule, alt, idx;
  bool operator==(const Pos& o) const { return rule == o.rule && alt == o.alt && idx == o.idx; }
  bool operator<(const Pos& o) const {
    if (rule != o.rule) return rule < o.rule;
    if (alt != o.alt) return alt < o.alt;
    return idx < o.idx;
  }
};
using Stack = std::vector<Pos>;
using StackSet = std::vector<Stack>;

constexpr size_t MAX_STACKS = 2048;
constexpr size_t MAX_DEPTH = 1024;

// ------------------------------------------------------------------------------------------ parser
struct Parser {
  const std::string& src;
  size_t p = 0;
  Grammar& g;
  std::map<std::string, int> names;
  std::string err;
  int helper = 0;
  int depth = 0;                 // open parentheses: parse_item / parse_alternates recurse once per level
  static constexpr int MAX_NESTING = 256;
  static constexpr size_t MAX_TEXT_BYTES = 64 * 1024;
  static constexpr size_t MAX_RULES = 20000;

  Parser(const std::string& s, Grammar& gr) : src(s), g(gr) {}

  bool fail(const std::string& what) {
    if (err.empty()) {
      size_t line = 1 + std::count(src.begin(), src.begin() + std::min(p, src.size()), '\n');
      err = what + " (line " + std::to_string(line) + ")";
    }
    return false;
  }
  void skip_ws(bool newlines) {
    while (p < src.size()) {
      char c = src[p];
      if (c == '#') {
        while (p < src.size() && src[p] != '\n') ++p;
      } else if (c == ' ' || c == '\t' || c == '\r' || (newlines && c == '\n')) {
        ++p;
      } else {
        break;
      }
    }
  }
  static bool name_start(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
  static bool name_char(char c) { return name_start(c) || (c >= '0' && c <= '9') || c == '-'; }
  std::string read_name() {
    size_t s = p;
    while (p < src.size() && name_char(src[p])) ++p;
    return src.substr(s, p - s);
  }
  int rule_id(const std::string& n) {
    auto it = names.find(n);
    if (it != names.end()) return it->second;
    int id = (int)g.rules.size();
    g.rules.push_back(Rule{n, {}});
    names[n] = id;
    return id;
  }
  int new_helper(const std::string& base) {
    int id = (int)g.rules.size();
    g.rules.push_back(Rule{base + "#" + std::to_string(helper++), {}});
    return id;
  }
  int add_set(const ByteSet& s) {
    g.sets.push_back(s);
    return (int)g.sets.size() - 1;
  }
  // one (possibly escaped) byte of a literal or class; returns -1 on error
  int read_byte() {
    if (p >= src.size()) return -1;
    unsigned char c = (unsigned char)src[p++];
    if (c != '\\') return c;
    if (p >= src.size()) return -1;
    char e = src[p++];
    switch (e) {
      case 'n': return '\n';
      case 't': return '\t';
      case 'r': return '\r';
      case '0': return 0;
      case 'x': {
        if (p + 2 > src.size()) return -1;
        int v = 0;
        for (int i = 0; i < 2; ++i) {
          char h = src[p++];
          int d = (h >= '0' && h <= '9') ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10
                  : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : -1;
          if (d < 0) return -1;
          v = v * 16 + d;
        }
        return v;
      }
      default: return (unsigned char)e;   // \\ \" \] \[ \- \^ ...
    }
  }

  // primary with postfix operators -> appended to `out`
  bool parse_item(const std::string& owner, Alt& out) {
    Alt prim;
    char c = src[p];
    if (c == '"') {
      ++p;
      while (true) {
        if (p >= src.size() || src[p] == '\n') return fail("unterminated literal");
        if (src[p] == '"') { ++p; break; }
        int b = read_byte();
        if (b < 0) return fail("bad escape in literal");
        ByteSet s{};
        bs_add(s, (unsigned)b);
        prim.push_back(Sym{-1, add_set(s)});
      }
    } else if (c == '[') {
      ++p;
      bool neg = false;
      if (p < src.size() && src[p] == '^') { neg = true; ++p; }
      ByteSet s{};
      bool first = true;
      while (true) {
        if (p >= src.size() || src[p] == '\n') return fail("unterminated character class");
        if (src[p] == ']' && !first) { ++p; break; }
        first = false;
        int lo = read_byte();
        if (lo < 0) return fail("bad escape in character class");
        if (lo >= 0x80) return fail("character classes hold ASCII only (non-ASCII text is matched through literals and negated classes)");
        int hi = lo;
        if (p + 1 < src.size() && src[p] == '-' && src[p + 1] != ']') {
          ++p;
          hi = read_byte();
          if (hi < 0 || hi >= 0x80) return fail("bad range in character class");
          if (hi < lo) return fail("reversed range in character class");
        }
        for (int b = lo; b <= hi; ++b) bs_add(s, (unsigned)b);
      }
      if (neg)
        for (auto& w : s) w = ~w;
      prim.push_back(Sym{-1, add_set(s)});
    } else if (c == '.') {
      ++p;
      ByteSet s;
      s.fill(~0ull);
      prim.push_back(Sym{-1, add_set(s)});
    } else if (c == '(') {
      ++p;
      if (++depth > MAX_NESTING) return fail("parentheses nested too deeply");
      int sub = new_helper(owner);
      if (!parse_alternates(owner, sub, true)) return false;
      --depth;
      skip_ws(true);
      if (p >= src.size() || src[p] != ')') return fail("expected )");
      ++p;
      prim.push_back(Sym{sub, -1});
    } else if (name_start(c)) {
      prim.push_back(Sym{rule_id(read_name()), -1});
    } else {
      return fail(std::string("unexpected character '") + c + "'");
    }
    // postfix operators apply to the whole primary (a multi-byte literal is one unit)
    skip_ws(false);
    while (p < src.size() && (src[p] == '*' || src[p] == '+' || src[p] == '?' || src[p] == '{')) {
      int lo = 0, hi = -1;      // hi < 0: unbounded
      char op = src[p++];
      if (op == '*') { lo = 0; hi = -1; }
      else if (op == '+') { lo = 1; hi = -1; }
      else if (op == '?') { lo = 0; hi = 1; }
      else {
        auto num = [&](int& v) {
          size_t s0 = p;
          v = 0;
          while (p < src.size() && src[p] >= '0' && src[p] <= '9') { v = v * 10 + (src[p++] - '0'); if (v > 10000) return false; }
          return p > s0;
        };
        if (!num(lo)) return fail("bad repetition count");
        hi = lo;
        if (p < src.size() && src[p] == ',') {
          ++p;
          if (p < src.size() && src[p] == '}') hi = -1;
          else if (!num(hi) || hi < lo) return fail("bad repetition count");
        }
        if (p >= src.size() || src[p] != '}') return fail("expected }");
        ++p;
        if (hi > 256 || lo > 256) return fail("repetition count above 256");
      }
      if (g.rules.size() > MAX_RULES) return fail("grammar too large");
      // unit ::= prim
      int unit = new_helper(owner);
      g.rules[unit].alts.push_back(prim);
      Alt rep;
      for (int i = 0; i < lo; ++i) rep.push_back(Sym{unit, -1});
      if (hi < 0) {                    // star ::= unit star | (empty)
        int star = new_helper(owner);
        g.rules[star].alts.push_back(Alt{Sym{unit, -1}, Sym{star, -1}});
        g.rules[star].alts.push_back(Alt{});
        rep.push_back(Sym{star, -1});
      } else if (hi > lo) {            // opt_k ::= unit opt_{k-1} | (empty)
        int next = -1;
        for (int i = 0; i < hi - lo; ++i) {
          int opt = new_helper(owner);
          Alt a{Sym{unit, -1}};
          if (next >= 0) a.push_back(Sym{next, -1});
          g.rules[opt].alts.push_back(a);
          g.rules[opt].alts.push_back(Alt{});
          next = opt;
        }
        rep.push_back(Sym{next, -1});
      }
      prim = rep;
      skip_ws(false);
    }
    out.insert(out.end(), prim.begin(), prim.end());
    return true;
  }

  // alternates of rule `id`; stops at ')' (nested) or at the start of the next rule / end of text
  bool parse_alternates(const std::string& owner, int id, bool nested) {
    Alt cur;
    while (true) {
      skip_ws(true);
      if (p >= src.size()) break;
      char c = src[p];
      if (c == ')') {
        if (!nested) return fail("unbalanced )");
        break;
      }
      if (c == '|') {
        ++p;
        g.rules[id].alts.push_back(cur);
        cur.clear();
        continue;
      }
      if (!nested && name_start(c)) {      // `name ::=` begins the next rule
        size_t save = p;
        read_name();
        skip_ws(false);
        bool is_def = src.compare(p, 3, "::=") == 0;
        p = save;
        if (is_def) break;
      }
      if (!parse_item(owner, cur)) return false;
    }
    g.rules[id].alts.push_back(cur);
    return true;
  }

  bool parse() {
    if (src.size() > MAX_TEXT_BYTES) return fail("grammar longer than 64 KiB");
    std::vector<char> defined;
    while (true) {
      skip_ws(true);
      if (p >= src.size()) break;
      if (!name_start(src[p])) return fail("expected a rule name");
      std::string n = read_name();
      skip_ws(false);
      if (src.compare(p, 3, "::=") != 0) return fail("expected ::= after rule name " + n);
      p += 3;
      int id = rule_id(n);
      if (defined.size() <= (size_t)id) defined.resize(g.rules.size() + 64, 0);
      if (defined[id]) return fail("rule " + n + " is defined twice");
      defined[id] = 1;
      if (!parse_alternates(n, id, false)) return false;
      if (g.rules.size() > MAX_RULES) return fail("grammar too large");
    }
    for (size_t i = 0; i < g.rules.size(); ++i)
      if (g.rules[i].alts.empty()) return fail("rule " + g.rules[i].name + " is used but never defined");
    auto it = names.find("root");
    if (it == names.end()) return fail("the grammar has no `root` rule");
    g.root = it->second;
    return true;
  }
};

// a rule that can reach itself as the leftmost symbol (through nullable prefixes) would expand forever
bool left_recursive(const Grammar& g, std::string& who) {
  const size_t n = g.rules.size();
  std::vector<char> nullable(n, 0);
  for (bool changed = true; changed;) {
    changed = false;
    for (size_t r = 0; r < n; ++r) {
      if (nullable[r]) continue;
      for (const Alt& a : g.rules[r].alts) {
        bool all = true;
        for (const Sym& s : a)
          if (s.rule < 0 || !nullable[s.rule]) { all = false; break; }
        if (all) { nullable[r] = 1; changed = true; break; }
      }
    }
  }
  std::vector<std::vector<int>> left(n);
  for (size_t r = 0; r < n; ++r)
    for (const Alt& a : g.rules[r].alts)
      for (const Sym& s : a) {
        if (s.rule < 0) break;
        left[r].push_back(s.rule);
        if (!nullable[s.rule]) break;
      }
  for (size_t r = 0; r < n; ++r) {
    std::vector<char> seen(n, 0);
    std::vector<int> todo(left[r]);
    while (!todo.empty()) {
      int x = todo.back();
      todo.pop_back();
      if ((size_t)x == r) { who = g.rules[r].name; return true; }
      if (seen[x]) continue;
      seen[x] = 1;
      for (int y : left[x]) todo.push_back(y);
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------ matcher
struct Matcher {
  const Grammar* g;

  // expand `st` until its innermost position is a byte set or it is empty; results appended to `out`
  bool expand(Stack st, StackSet& out, size_t depth) const {
    if (depth > MAX_DEPTH || out.size() > MAX_STACKS) return false;
    while (!st.empty()) {
      const Pos top = st.back();
      const Alt& a = g->rules[top.rule].alts[top.alt];
      if (top.idx >= (int)a.size()) { st.pop_back(); continue; }     // alternative finished
      const Sym& s = a[top.idx];
      if (s.rule < 0) { out.push_back(std::move(st)); return true; }
      // step past the reference in the caller (dropping the caller when that was its last symbol),
      // then fork one stack per alternative of the callee
      st.pop_back();
      if (top.idx + 1 < (int)a.size()) st.push_back(Pos{top.rule, top.alt, top.idx + 1});
      const Rule& callee = g->rules[s.rule];
      for (size_t k = 0; k < callee.alts.size(); ++k) {
        Stack next = st;
        next.push_back(Pos{s.rule, (int)k, 0});
        if (!expand(std::move(next), out, depth + 1)) return false;
      }
      return true;
    }
    out.push_back(std::move(st));     // empty: a complete sentence
    return true;
  }

  static void dedup(StackSet& s) {
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
  }

  bool start(StackSet& out) const {
    out.clear();
    const Rule& r = g->rules[g->root];
    for (size_t k = 0; k < r.alts.size(); ++k)
      if (!expand(Stack{Pos{g->root, (int)k, 0}}, out, 0)) return false;
    dedup(out);
    return true;
  }

  // stacks after byte c; false only on overflow (an empty result means "rejected")
  bool accept(const StackSet& in, unsigned c, StackSet& out) const {
    out.clear();
    for (const Stack& st : in) {
      if (st.empty()) continue;
      const Pos top = st.back();
      const Alt& a = g->rules[top.rule].alts[top.alt];
      if (!bs_has(g->sets[a[top.idx].set], c)) continue;
      Stack next = st;
      next.pop_back();
      if (top.idx + 1 < (int)a.size()) next.push_back(Pos{top.rule, top.alt, top.idx + 1});
      if (!expand(std::move(next), out, 0)) return false;
    }
    if (out.size() > 1) dedup(out);
    return true;
  }

  ByteSet first_bytes(const StackSet& in) const {
    ByteSet u{};
    for (const Stack& st : in) {
      if (st.empty()) continue;
      const Pos top = st.back();
      const ByteSet& s = g->sets[g->rules[top.rule].alts[top.alt][top.idx].set];
      for (int i = 0; i < 4; ++i) u[i] |= s[i];
    }
    return u;
  }
};

// Lazily built DFA over interned stack sets, shared by every request that uses the grammar.
struct Lazy {
  std::mutex mu;
  std::map<StackSet, int> ids;
  std::vector<const StackSet*> sets;                 // id -> key of `ids` (map nodes do not move)
  std::unordered_map<uint64_t, int> trans;           // (id << 8 | byte) -> id, or -1: rejected
  static constexpr size_t MAX_SETS = 50000;

  int intern(const StackSet& s) {
    auto it = ids.find(s);
    if (it != ids.end()) return it->second;
    const int id = (int)sets.size();
    auto ins = ids.emplace(s, id);
    sets.push_back(&ins.first->first);
    return id;
  }
  int step(const Matcher& m, int id, unsigned c) {
    const uint64_t key = ((uint64_t)id << 8) | c;
    auto it = trans.find(key);
    if (it != trans.end()) return it->second;
    StackSet next;
    int to = -1;
    if (m.accept(*sets[id], c, next) && !next.empty()) to = intern(next);
    trans.emplace(key, to);
    return to;
  }
  void trim() {
    if (sets.size() <= MAX_SETS) return;
    ids.clear();
    sets.clear();
    trans.clear();
  }
};

struct GrammarBox {
  Grammar g;
  Lazy lazy;
};

struct State {
  std::shared_ptr<GrammarBox> box;
  StackSet stacks;
};

// ------------------------------------------------------------------------------------------ vocabulary trie
struct Trie {
  struct Node {
    std::vector<std::pair<unsigned char, int>> kids;   // sorted by byte
    std::vector<int> tokens;                           // token ids spelled by the path to this node
  };
  std::vector<Node> nodes;
  long n_tokens = 0;
};

void walk(const Matcher& m, Lazy& lazy, const Trie& t, int node, int set_id, uint8_t* out, long& allowed) {
  for (const auto& kid : t.nodes[node].kids) {
    const int to = lazy.step(m, set_id, kid.first);
    if (to < 0) continue;
    const Trie::Node& child = t.nodes[kid.second];
    for (int id : child.tokens) {
      if (!out[id]) ++allowed;
      out[id] = 1;
    }
    if (!child.kids.empty()) walk(m, lazy, t, kid.second, to, out, allowed);
  }
}

}  // namespace

extern "C" {

// Compiled grammar handle, or null with a message in `err`.
void* vdamd_cfg_compile(const char* text, long n, char* err, int errlen) {
  auto box = std::make_shared<GrammarBox>();
  Grammar* g = &box->g;
  std::string src(text, (size_t)n);
  Parser ps(src, *g);
  std::string msg;
  if (!ps.parse()) msg = ps.err;
  std::string who;
  if (msg.empty() && left_recursive(*g, who)) msg = "rule " + who + " is left-recursive (rewrite it with a repetition)";
  if (msg.empty()) {
    StackSet s;
    Matcher m{g};
    if (!m.start(s)) msg = "the grammar is too ambiguous to expand";
  }
  if (!msg.empty()) {
    if (err != nullptr && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
    return nullptr;
  }
  return new std::shared_ptr<GrammarBox>(box);
}

void vdamd_cfg_free(void* h) { delete static_cast<std::shared_ptr<GrammarBox>*>(h); }

void* vdamd_cfg_state_new(void* h) {
  auto* st = new State{*static_cast<std::shared_ptr<GrammarBox>*>(h), {}};
  Matcher m{&st->box->g};
  m.start(st->stacks);
  return st;
}

void vdamd_cfg_state_free(void* s) { delete static_cast<State*>(s); }

// 0: consumed (state advanced); -1: the bytes do not continue any sentence (state unchanged)
int vdamd_cfg_advance(void* s, const char* data, long n) {
  State* st = static_cast<State*>(s);
  Matcher m{&st->box->g};
  StackSet cur = st->stacks, next;
  for (long i = 0; i < n; ++i) {
    if (!m.accept(cur, (unsigned char)data[i], next) || next.empty()) return -1;
    cur.swap(next);
  }
  st->stacks.swap(cur);
  return 0;
}

// bit 0: the text so far is a complete sentence; bit 1: some longer sentence extends it
int vdamd_cfg_flags(void* s) {
  const State* st = static_cast<State*>(s);
  int f = 0;
  for (const Stack& x : st->stacks) f |= x.empty() ? 1 : 2;
  return f;
}

// a byte string that identifies the matcher position (for caching masks); returns its length
long vdamd_cfg_state_key(void* s, char* buf, long cap) {
  const State* st = static_cast<State*>(s);
  long need = 0;
  for (const Stack& x : st->stacks) need += 4 + (long)x.size() * 12;
  if (buf == nullptr || need > cap) return need;
  char* p = buf;
  for (const Stack& x : st->stacks) {
    int32_t n = (int32_t)x.size();
    std::memcpy(p, &n, 4);
    p += 4;
    for (const Pos& q : x) {
      int32_t v[3] = {q.rule, q.alt, q.idx};
      std::memcpy(p, v, 12);
      p += 12;
    }
  }
  return need;
}

// Vocabulary trie over `n` tokens (bytes blob[offsets[i] .. offsets[i+1])); empty tokens are left out.
void* vdamd_trie_build(const uint8_t* blob, const int64_t* offsets, long n) {
  Trie* t = new Trie();
  t->n_tokens = n;
  t->nodes.emplace_back();
  for (long i = 0; i < n; ++i) {
    const int64_t a = offsets[i], b = offsets[i + 1];
    if (b <= a) continue;
    int node = 0;
    for (int64_t k = a; k < b; ++k) {
      const unsigned char c = blob[k];
      auto& kids = t->nodes[node].kids;
      auto it = std::lower_bound(kids.begin(), kids.end(), std::make_pair(c, -1),
                                 [](const std::pair<unsigned char, int>& x, const std::pair<unsigned char, int>& y) {
                                   return x.first < y.first;
                                 });
      if (it != kids.end() && it->first == c) {
        node = it->second;
      } else {
        const int fresh = (int)t->nodes.size();
        kids.insert(it, std::make_pair(c, fresh));   // (invalidates nothing we still hold: node is an index)
        t->nodes.emplace_back();
        node = fresh;
      }
    }
    t->nodes[node].tokens.push_back((int)i);
  }
  return t;
}

void vdamd_trie_free(void* t) { delete static_cast<Trie*>(t); }

// out[i] = 1 for every token whose bytes the grammar accepts next; returns how many
long vdamd_cfg_mask(void* s, void* trie, uint8_t* out) {
  const State* st = static_cast<State*>(s);
  const Trie* t = static_cast<Trie*>(trie);
  std::memset(out, 0, (size_t)t->n_tokens);
  long allowed = 0;
  Matcher m{&st->box->g};
  Lazy& lazy = st->box->lazy;
  std::lock_guard<std::mutex> hold(lazy.mu);
  lazy.trim();
  walk(m, lazy, *t, 0, lazy.intern(st->stacks), out, allowed);
  return allowed;
}

}  // extern "C"
