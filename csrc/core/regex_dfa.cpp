// Regular expression -> byte-level DFA, for constrained decoding (guided_regex, and JSON schemas
// that the Python side lowers to a regular expression).
//
// The whole output of a request must match the expression (full match).  The expression is parsed
// to an AST, lowered to a Thompson NFA, determinised by subset construction, and then trimmed: a
// DFA state from which no accepting state can be reached is dead, and a transition into it is
// removed.  After trimming, "token t may come next" is exactly "walking t's bytes from the current
// state never leaves the DFA", which is what the mask loop evaluates for every vocabulary entry.
//
// Syntax: literals (UTF-8 passes through byte by byte), `.`, escapes \d \D \w \W \s \S \n \r \t \f \v
// \xHH \uHHHH and escaped punctuation, classes [a-z0-9_] / [^...] (with the escapes above), groups
// (...) and (?:...), alternation, quantifiers * + ? {m} {m,} {m,n}; ^ and $ are accepted and
// ignored (the match is anchored anyway).  `.` and negated classes also admit bytes >= 0x80, so
// non-ASCII text is allowed wherever an "anything but" is.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

typedef std::vector<bool> ByteSet;   // 256 entries

struct Node {
  enum Kind { SET, CAT, ALT, REP, EMPTY } kind = EMPTY;
  ByteSet set;                          // SET
  std::vector<std::unique_ptr<Node>> kids;   // CAT / ALT / REP (one child)
  int lo = 0, hi = -1;                  // REP: hi < 0 = unbounded
};
typedef std::unique_ptr<Node> NodeP;

struct Parser {
  const unsigned char* p;
  const unsigned char* e;
  long budget = 400000;                 // NFA positions the expansion of counted repeats may create
  int depth = 0;                        // open groups: the parser and the NFA builder recurse once per level
  static constexpr int MAX_DEPTH = 256;
  static constexpr int MAX_STACKED_QUANTIFIERS = 4;

  [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error(what); }
  bool more() const { return p < e; }

  static NodeP mk(Node::Kind k) { NodeP n(new Node); n->kind = k; return n; }
  static NodeP lit(unsigned char c) {
    NodeP n = mk(Node::SET);
    n->set.assign(256, false);
    n->set[c] = true;
    return n;
  }
  static void add_range(ByteSet& s, int a, int b) { for (int c = a; c <= b; ++c) s[c] = true; }
  static void add_class(ByteSet& s, char c) {
    ByteSet t(256, false);
    switch (c | 0x20) {
      case 'd': add_range(t, '0', '9'); break;
      case 'w': add_range(t, '0', '9'); add_range(t, 'a', 'z'); add_range(t, 'A', 'Z'); t['_'] = true; break;
      case 's': t[' '] = t['\t'] = t['\n'] = t['\r'] = t['\f'] = t['\v'] = true; break;
    }
    const bool neg = c >= 'A' && c <= 'Z';
    for (int i = 0; i < 256; ++i) if (t[i] != neg) s[i] = true;
  }
  int hexval(int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) {
      if (!more()) fail("truncated hex escape");
      const int c = *p++;
      int d = c >= '0' && c <= '9' ? c - '0' : (c | 0x20) >= 'a' && (c | 0x20) <= 'f' ? (c | 0x20) - 'a' + 10 : -1;
      if (d < 0) fail("bad hex escape");
      v = v * 16 + d;
    }
    return v;
  }
  static std::string utf8(int cp) {
    std::string s;
    if (cp < 0x80) s += (char)cp;
    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
    else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    return s;
  }
  // an escape after the backslash: a class (returns true, fills set) or a byte string
  bool escape(ByteSet& set, std::string& bytes) {
    if (!more()) fail("trailing backslash");
    const unsigned char c = *p++;
    switch (c) {
      case 'd': case 'D': case 'w': case 'W': case 's': case 'S':
        set.assign(256, false);
        add_class(set, (char)c);
        return true;
      case 'n': bytes = "\n"; return false;
      case 'r': bytes = "\r"; return false;
      case 't': bytes = "\t"; return false;
      case 'f': bytes = "\f"; return false;
      case 'v': bytes = "\v"; return false;
      case '0': bytes = std::string(1, '\0'); return false;
      case 'x': bytes = std::string(1, (char)hexval(2)); return false;
      case 'u': bytes = utf8(hexval(4)); return false;
      default:
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '1' && c <= '9'))
          fail(std::string("unsupported escape \\") + (char)c);
        bytes = std::string(1, (char)c);
        return false;
    }
  }
  NodeP bytes_node(const std::string& s) {
    if (s.size() == 1) return lit((unsigned char)s[0]);
    NodeP n = mk(Node::CAT);
    for (unsigned char c : s) n->kids.push_back(lit(c));
    return n;
  }
  NodeP char_class() {                  // after '['
    NodeP n = mk(Node::SET);
    n->set.assign(256, false);
    bool neg = false;
    if (more() && *p == '^') { neg = true; ++p; }
    bool first = true;
    while (true) {
      if (!more()) fail("unterminated [");
      unsigned char c = *p++;
      if (c == ']' && !first) break;
      first = false;
      int lo;
      if (c == '\\') {
        ByteSet cs; std::string bs;
        if (escape(cs, bs)) { for (int i = 0; i < 256; ++i) if (cs[i]) n->set[i] = true; continue; }
        if (bs.size() != 1) fail("multi-byte escape inside a class");
        lo = (unsigned char)bs[0];
      } else {
        if (c >= 0x80) fail("non-ASCII characters inside a class are not supported");
        lo = c;
      }
      int hi = lo;
      if (p + 1 < e && *p == '-' && p[1] != ']') {
        ++p;
        unsigned char d = *p++;
        if (d == '\\') {
          ByteSet cs; std::string bs;
          if (escape(cs, bs) || bs.size() != 1) fail("bad range end");
          hi = (unsigned char)bs[0];
        } else hi = d;
        if (hi < lo) fail("reversed range");
      }
      add_range(n->set, lo, hi);
    }
    if (neg) for (int i = 0; i < 256; ++i) n->set[i] = !n->set[i];
    return n;
  }
  NodeP atom() {
    const unsigned char c = *p++;
    switch (c) {
      case '(': {
        if (p + 1 < e && p[0] == '?' && p[1] == ':') p += 2;
        else if (more() && *p == '?') fail("only (?:...) groups are supported");
        if (++depth > MAX_DEPTH) fail("groups nested too deeply");
        NodeP n = alt();
        --depth;
        if (!more() || *p != ')') fail("missing )");
        ++p;
        return n;
      }
      case '[': return char_class();
      case '.': {
        NodeP n = mk(Node::SET);
        n->set.assign(256, true);
        n->set['\n'] = false;
        return n;
      }
      case '\\': {
        ByteSet cs; std::string bs;
        if (escape(cs, bs)) { NodeP n = mk(Node::SET); n->set = cs; return n; }
        return bytes_node(bs);
      }
      case '^': case '$': return mk(Node::EMPTY);
      case '*': case '+': case '?': fail("quantifier without an operand");
      default: return lit(c);
    }
  }
  long size_of(const Node* n) const {
    switch (n->kind) {
      // what Nfa::build allocates, roughly: an empty operand and every repetition's fork cost states too
      case Node::SET: return 1;
      case Node::EMPTY: return 1;
      case Node::REP: {
        const long s = size_of(n->kids[0].get()) + 1;
        const long reps = n->hi < 0 ? n->lo + 1 : n->hi;
        return s * (reps < 1 ? 1 : reps);
      }
      default: { long s = 0; for (auto& k : n->kids) s += size_of(k.get()); return s; }
    }
  }
  NodeP repeat() {
    NodeP a = atom();
    int stacked = 0;
    while (more()) {
      int lo, hi;
      const unsigned char c = *p;
      if (c == '*') { lo = 0; hi = -1; ++p; }
      else if (c == '+') { lo = 1; hi = -1; ++p; }
      else if (c == '?') { lo = 0; hi = 1; ++p; }
      else if (c == '{') {
        const unsigned char* q = p + 1;
        long a1 = 0, b1 = -1; bool dig = false;
        while (q < e && *q >= '0' && *q <= '9') { a1 = a1 * 10 + (*q++ - '0'); dig = true; if (a1 > 100000) fail("repeat count too large"); }
        if (!dig) break;                      // a literal '{'
        if (q < e && *q == ',') {
          ++q;
          if (q < e && *q >= '0' && *q <= '9') {
            b1 = 0;
            while (q < e && *q >= '0' && *q <= '9') { b1 = b1 * 10 + (*q++ - '0'); if (b1 > 100000) fail("repeat count too large"); }
          }
        } else b1 = a1;
        if (q >= e || *q != '}') break;       // literal
        if (b1 >= 0 && b1 < a1) fail("reversed repeat bounds");
        lo = (int)a1; hi = (int)b1; p = q + 1;
      } else break;
      if (more() && (*p == '?' || *p == '+')) ++p;   // lazy / possessive markers do not change the language
      if (++stacked > MAX_STACKED_QUANTIFIERS) fail("too many quantifiers on one operand");
      NodeP r = mk(Node::REP);
      r->lo = lo; r->hi = hi;
      r->kids.push_back(std::move(a));
      budget -= size_of(r.get());
      if (budget < 0) fail("expression too large after expanding counted repeats");
      a = std::move(r);
    }
    return a;
  }
  NodeP cat() {
    NodeP n = mk(Node::CAT);
    while (more() && *p != '|' && *p != ')') n->kids.push_back(repeat());
    if (n->kids.empty()) return mk(Node::EMPTY);
    if (n->kids.size() == 1) return std::move(n->kids[0]);
    return n;
  }
  NodeP alt() {
    NodeP first = cat();
    if (!more() || *p != '|') return first;
    NodeP n = mk(Node::ALT);
    n->kids.push_back(std::move(first));
    while (more() && *p == '|') { ++p; n->kids.push_back(cat()); }
    return n;
  }
};

// ---- Thompson NFA: state i has either a byte set + one target, or up to two epsilon targets
struct Nfa {
  struct St { int set = -1; int to = -1; int e1 = -1, e2 = -1; };
  std::vector<St> st;
  std::vector<ByteSet> sets;
  int add() {
    if (st.size() >= 2000000) throw std::runtime_error("expression too large");
    st.emplace_back();
    return (int)st.size() - 1;
  }
  // builds n between a fresh start and a fresh end; returns (start, end)
  std::pair<int, int> build(const Node* n) {
    switch (n->kind) {
      case Node::EMPTY: { const int s = add(), t = add(); st[s].e1 = t; return {s, t}; }
      case Node::SET: {
        const int s = add(), t = add();
        sets.push_back(n->set);
        st[s].set = (int)sets.size() - 1; st[s].to = t;
        return {s, t};
      }
      case Node::CAT: {
        std::pair<int, int> all = build(n->kids[0].get());
        for (size_t i = 1; i < n->kids.size(); ++i) {
          auto nx = build(n->kids[i].get());
          st[all.second].e1 = nx.first;
          all.second = nx.second;
        }
        return all;
      }
      case Node::ALT: {
        const int s = add(), t = add();
        int cur = s;
        for (size_t i = 0; i < n->kids.size(); ++i) {
          auto k = build(n->kids[i].get());
          st[k.second].e1 = t;
          if (i + 1 < n->kids.size()) { const int nxt = add(); st[cur].e1 = k.first; st[cur].e2 = nxt; cur = nxt; }
          else st[cur].e1 = k.first;
        }
        return {s, t};
      }
      case Node::REP: {
        const Node* k = n->kids[0].get();
        const int s = add();
        int cur = s;
        for (int i = 0; i < n->lo; ++i) { auto c = build(k); st[cur].e1 = c.first; cur = c.second; }
        const int t = add();
        if (n->hi < 0) {                      // k*
          const int loop = add();
          st[cur].e1 = loop;
          auto c = build(k);
          st[loop].e1 = c.first; st[loop].e2 = t;
          st[c.second].e1 = loop;
        } else {
          for (int i = n->lo; i < n->hi; ++i) {   // (k (k (k)?)?)?
            auto c = build(k);
            const int fork = add();
            st[cur].e1 = fork;
            st[fork].e1 = c.first; st[fork].e2 = t;
            cur = c.second;
          }
          st[cur].e1 = t;
        }
        return {s, t};
      }
    }
    return {0, 0};
  }
};

struct Dfa {
  std::vector<int32_t> trans;          // [state][256], -1 = no transition
  std::vector<uint8_t> accepting, live_out;
  int start = 0;
  int n() const { return (int)accepting.size(); }
};

// `set` grows to its epsilon closure; stamp[s] == gen marks membership (no per-call clearing)
void closure(const Nfa& nfa, std::vector<int>& set, std::vector<int>& stamp, int gen) {
  std::vector<int> stack(set);
  while (!stack.empty()) {
    const int s = stack.back(); stack.pop_back();
    for (int t : {nfa.st[s].e1, nfa.st[s].e2})
      if (t >= 0 && stamp[t] != gen) { stamp[t] = gen; set.push_back(t); stack.push_back(t); }
  }
}

constexpr long MAX_PATTERN_BYTES = 64 * 1024;
constexpr long MAX_SUBSET_WORK = 400L * 1000 * 1000;   // NFA states visited while determinising

Dfa* compile(const unsigned char* pat, long len, int max_states) {
  if (len > MAX_PATTERN_BYTES) throw std::runtime_error("expression longer than 64 KiB");
  Parser ps{pat, pat + len};
  NodeP root = ps.alt();
  if (ps.more()) ps.fail(*ps.p == ')' ? "unbalanced )" : "trailing input");
  Nfa nfa;
  auto se = nfa.build(root.get());
  const int nn = (int)nfa.st.size();
  std::map<std::vector<int>, int> ids;
  std::vector<std::vector<int>> members;
  std::vector<int32_t> trans;
  std::vector<int> stamp(nn, 0);
  int gen = 0;
  long work = 0;
  auto intern = [&](std::vector<int>& raw) -> int {
    ++gen;
    for (int s : raw) stamp[s] = gen;
    closure(nfa, raw, stamp, gen);
    work += (long)raw.size();
    if (work > MAX_SUBSET_WORK) throw std::runtime_error("expression too expensive to determinise");
    std::vector<int> key;
    for (int s : raw)
      if (nfa.st[s].set >= 0 || s == se.second) key.push_back(s);   // only states that matter
    std::sort(key.begin(), key.end());
    key.erase(std::unique(key.begin(), key.end()), key.end());
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    if ((int)members.size() >= max_states) throw std::runtime_error("expression needs too many DFA states");
    const int id = (int)members.size();
    ids.emplace(key, id);
    members.push_back(key);
    trans.resize((size_t)(id + 1) * 256, -1);
    return id;
  };
  std::vector<int> init{se.first};
  intern(init);
  for (int d = 0; d < (int)members.size(); ++d) {
    // group bytes by the set of NFA targets they reach
    std::map<std::vector<int>, std::vector<int>> by_target;
    const std::vector<int> mem = members[d];
    work += 256L * (long)mem.size();
    if (work > MAX_SUBSET_WORK) throw std::runtime_error("expression too expensive to determinise");
    for (int c = 0; c < 256; ++c) {
      std::vector<int> tg;
      for (int s : mem)
        if (nfa.st[s].set >= 0 && nfa.sets[nfa.st[s].set][c]) tg.push_back(nfa.st[s].to);
      if (!tg.empty()) by_target[tg].push_back(c);
    }
    for (auto& kv : by_target) {
      std::vector<int> raw = kv.first;
      const int to = intern(raw);
      for (int c : kv.second) trans[(size_t)d * 256 + c] = to;
    }
  }
  const int nd = (int)members.size();
  std::unique_ptr<Dfa> dfa(new Dfa);
  dfa->accepting.assign(nd, 0);
  for (int d = 0; d < nd; ++d)
    for (int s : members[d]) if (s == se.second) dfa->accepting[d] = 1;
  // trim: keep only states that can reach an accepting state
  std::vector<std::vector<int>> rev(nd);
  for (int d = 0; d < nd; ++d)
    for (int c = 0; c < 256; ++c) { const int t = trans[(size_t)d * 256 + c]; if (t >= 0) rev[t].push_back(d); }
  std::vector<char> good(nd, 0);
  std::vector<int> stack;
  for (int d = 0; d < nd; ++d) if (dfa->accepting[d]) { good[d] = 1; stack.push_back(d); }
  while (!stack.empty()) {
    const int t = stack.back(); stack.pop_back();
    for (int d : rev[t]) if (!good[d]) { good[d] = 1; stack.push_back(d); }
  }
  if (!good[0]) throw std::runtime_error("expression matches nothing");
  dfa->live_out.assign(nd, 0);
  for (int d = 0; d < nd; ++d)
    for (int c = 0; c < 256; ++c) {
      int32_t& t = trans[(size_t)d * 256 + c];
      if (t >= 0 && !good[t]) t = -1;
      if (t >= 0) dfa->live_out[d] = 1;
    }
  dfa->trans.swap(trans);
  return dfa.release();
}

}  // namespace

extern "C" {

__attribute__((visibility("default")))
void* vdamd_dfa_compile(const char* pattern, long len, int max_states, char* err, int errlen) {
  try {
    return compile(reinterpret_cast<const unsigned char*>(pattern), len, max_states);
  } catch (const std::exception& ex) {
    if (err && errlen > 0) { std::strncpy(err, ex.what(), errlen - 1); err[errlen - 1] = 0; }
    return nullptr;
  }
}

__attribute__((visibility("default"))) void vdamd_dfa_free(void* h) { delete static_cast<Dfa*>(h); }
__attribute__((visibility("default"))) int vdamd_dfa_num_states(const void* h) { return static_cast<const Dfa*>(h)->n(); }

// state after consuming data from `state`, or -1 when the text can no longer match
__attribute__((visibility("default")))
int vdamd_dfa_step(const void* h, int state, const unsigned char* data, long n) {
  const Dfa* d = static_cast<const Dfa*>(h);
  if (state < 0 || state >= d->n()) return -1;
  for (long i = 0; i < n && state >= 0; ++i) state = d->trans[(size_t)state * 256 + data[i]];
  return state;
}

// bit 0: the text so far is a complete match; bit 1: it can be extended
__attribute__((visibility("default")))
int vdamd_dfa_flags(const void* h, int state) {
  const Dfa* d = static_cast<const Dfa*>(h);
  if (state < 0 || state >= d->n()) return 0;
  return (d->accepting[state] ? 1 : 0) | (d->live_out[state] ? 2 : 0);
}

// mask[i] = 1 when token i's bytes can follow from `state`; empty tokens are never allowed
__attribute__((visibility("default")))
long vdamd_dfa_mask(const void* h, int state, const unsigned char* blob, const long* offsets, long n_tokens,
                    unsigned char* mask) {
  const Dfa* d = static_cast<const Dfa*>(h);
  long allowed = 0;
  const bool valid = state >= 0 && state < d->n();
  for (long i = 0; i < n_tokens; ++i) {
    const long b = offsets[i], e = offsets[i + 1];
    int s = valid && e > b ? state : -1;
    for (long j = b; j < e && s >= 0; ++j) s = d->trans[(size_t)s * 256 + blob[j]];
    mask[i] = s >= 0 ? 1 : 0;
    allowed += s >= 0;
  }
  return allowed;
}

}  // extern "C"
