// Byte-level JSON automaton for constrained decoding (response_format = json_object).
//
// The sampler asks, before every token of a constrained request, which vocabulary
// entries keep the text generated so far a prefix of one JSON object.  That is a
// walk of this automaton over each candidate token's bytes from the current
// state: ~150k tokens x a few bytes, a couple of milliseconds here and far too
// slow in the interpreter.  The state is a small POD (mode + a nesting stack of at
// most 64 levels), so trying a token is a struct copy and a short loop.
//
// Grammar: RFC 8259 with the top-level value restricted to an object (what the
// OpenAI "json_object" mode promises).  Bytes >= 0x80 pass through inside strings
// (tokens may split a multi-byte character, so UTF-8 is not validated here);
// control bytes < 0x20 are rejected inside strings; runs of insignificant
// whitespace are capped at 64 bytes so a model cannot pad forever.
#include <cstdint>
#include <cstring>

namespace {

constexpr int MAX_DEPTH = 64;
constexpr int MAX_WS_RUN = 64;

enum Mode : uint8_t {
  TOP,          // nothing yet: whitespace or '{'
  VALUE,        // a value must start
  OBJ_OPEN,     // after '{': '"' or '}'
  OBJ_KEY,      // after ',' in an object: '"'
  OBJ_COLON,    // after a key: ':'
  ARR_OPEN,     // after '[': a value or ']'
  AFTER_VALUE,  // a value just ended: ',' or the closer of the enclosing container
  STR, STR_ESC, STR_U,
  NUM_MINUS, NUM_ZERO, NUM_INT, NUM_DOT, NUM_FRAC, NUM_E, NUM_ESIGN, NUM_EXP,
  LIT,
};

struct State {
  uint8_t mode;
  uint8_t is_key;     // the string being read is an object key
  uint8_t aux;        // STR_U: hex digits left; LIT: position in the literal
  uint8_t lit;        // LIT: 0 true, 1 false, 2 null
  uint8_t depth;
  uint8_t ws_run;
  char stack[MAX_DEPTH];      // entries above depth stay zero, so equal positions have equal bytes (mask caching)
};

const char* const LITERALS[3] = {"true", "false", "null"};

inline bool is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
inline bool is_hex(unsigned char c) {
  return is_digit(c) || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F');
}

inline bool begin_value(State& s, unsigned char c) {
  switch (c) {
    case '{':
      if (s.depth >= MAX_DEPTH) return false;
      s.stack[s.depth++] = '{';
      s.mode = OBJ_OPEN;
      return true;
    case '[':
      if (s.depth >= MAX_DEPTH) return false;
      s.stack[s.depth++] = '[';
      s.mode = ARR_OPEN;
      return true;
    case '"': s.mode = STR; s.is_key = 0; return true;
    case '-': s.mode = NUM_MINUS; return true;
    case '0': s.mode = NUM_ZERO; return true;
    case 't': s.mode = LIT; s.lit = 0; s.aux = 1; return true;
    case 'f': s.mode = LIT; s.lit = 1; s.aux = 1; return true;
    case 'n': s.mode = LIT; s.lit = 2; s.aux = 1; return true;
    default:
      if (c >= '1' && c <= '9') { s.mode = NUM_INT; return true; }
      return false;
  }
}

bool step(State& s, unsigned char c);

// a number ends at the first byte that cannot continue it; that byte then belongs to the enclosing container
inline bool end_number(State& s, unsigned char c) {
  s.mode = AFTER_VALUE;
  return step(s, c);
}

bool step(State& s, unsigned char c) {
  const uint8_t m = s.mode;
  if (m != STR && m != STR_ESC && m != STR_U) {
    if (is_ws(c)) {
      // whitespace ends a number and is insignificant elsewhere (never inside a literal or a half-read number)
      if (m == LIT || m == NUM_MINUS || m == NUM_DOT || m == NUM_E || m == NUM_ESIGN) return false;
      if (m == NUM_ZERO || m == NUM_INT || m == NUM_FRAC || m == NUM_EXP) s.mode = AFTER_VALUE;
      return ++s.ws_run <= MAX_WS_RUN;
    }
    s.ws_run = 0;
  }
  switch (m) {
    case TOP:
      if (c != '{') return false;
      return begin_value(s, c);
    case VALUE: return begin_value(s, c);
    case OBJ_OPEN:
      if (c == '}') { s.stack[--s.depth] = 0; s.mode = AFTER_VALUE; return true; }
      if (c == '"') { s.mode = STR; s.is_key = 1; return true; }
      return false;
    case OBJ_KEY:
      if (c == '"') { s.mode = STR; s.is_key = 1; return true; }
      return false;
    case OBJ_COLON:
      if (c == ':') { s.mode = VALUE; return true; }
      return false;
    case ARR_OPEN:
      if (c == ']') { s.stack[--s.depth] = 0; s.mode = AFTER_VALUE; return true; }
      return begin_value(s, c);
    case AFTER_VALUE:
      if (s.depth == 0) return false;                 // the object is complete: only whitespace may follow
      if (s.stack[s.depth - 1] == '{') {
        if (c == ',') { s.mode = OBJ_KEY; return true; }
        if (c == '}') { s.stack[--s.depth] = 0; return true; }
      } else {
        if (c == ',') { s.mode = VALUE; return true; }
        if (c == ']') { s.stack[--s.depth] = 0; return true; }
      }
      return false;
    case STR:
      if (c == '"') { s.mode = s.is_key ? OBJ_COLON : AFTER_VALUE; s.is_key = 0; return true; }
      if (c == '\\') { s.mode = STR_ESC; return true; }
      return c >= 0x20;
    case STR_ESC:
      if (c == 'u') { s.mode = STR_U; s.aux = 4; return true; }
      if (std::strchr("\"\\/bfnrt", c) != nullptr && c != 0) { s.mode = STR; return true; }
      return false;
    case STR_U:
      if (!is_hex(c)) return false;
      if (--s.aux == 0) s.mode = STR;
      return true;
    case NUM_MINUS:
      if (c == '0') { s.mode = NUM_ZERO; return true; }
      if (c >= '1' && c <= '9') { s.mode = NUM_INT; return true; }
      return false;
    case NUM_ZERO:
      if (c == '.') { s.mode = NUM_DOT; return true; }
      if (c == 'e' || c == 'E') { s.mode = NUM_E; return true; }
      if (is_digit(c)) return false;                  // no leading zeros
      return end_number(s, c);
    case NUM_INT:
      if (is_digit(c)) return true;
      if (c == '.') { s.mode = NUM_DOT; return true; }
      if (c == 'e' || c == 'E') { s.mode = NUM_E; return true; }
      return end_number(s, c);
    case NUM_DOT:
      if (is_digit(c)) { s.mode = NUM_FRAC; return true; }
      return false;
    case NUM_FRAC:
      if (is_digit(c)) return true;
      if (c == 'e' || c == 'E') { s.mode = NUM_E; return true; }
      return end_number(s, c);
    case NUM_E:
      if (c == '+' || c == '-') { s.mode = NUM_ESIGN; return true; }
      if (is_digit(c)) { s.mode = NUM_EXP; return true; }
      return false;
    case NUM_ESIGN:
      if (is_digit(c)) { s.mode = NUM_EXP; return true; }
      return false;
    case NUM_EXP:
      if (is_digit(c)) return true;
      return end_number(s, c);
    case LIT: {
      const char* lit = LITERALS[s.lit];
      if ((unsigned char)lit[s.aux] != c) return false;
      if (lit[++s.aux] == 0) s.mode = AFTER_VALUE;
      return true;
    }
  }
  return false;
}

inline bool feed(State& s, const unsigned char* p, long n) {
  for (long i = 0; i < n; ++i)
    if (!step(s, p[i])) return false;
  return true;
}

}  // namespace

extern "C" {

int vdamd_json_state_bytes() { return (int)sizeof(State); }

void vdamd_json_reset(void* state) { std::memset(state, 0, sizeof(State)); }

// 0 when every byte was accepted (the state advanced); -1 otherwise (the state is left untouched)
int vdamd_json_advance(void* state, const unsigned char* bytes, long n) {
  State tmp = *static_cast<State*>(state);
  if (!feed(tmp, bytes, n)) return -1;
  *static_cast<State*>(state) = tmp;
  return 0;
}

// 1 once the top-level object has closed
int vdamd_json_done(const void* state) {
  const State* s = static_cast<const State*>(state);
  return s->mode == AFTER_VALUE && s->depth == 0;
}

// mask[i] = 1 when token i (bytes blob[offsets[i] .. offsets[i+1])) keeps the text a JSON prefix; tokens with
// no bytes (special tokens) are never allowed.  Returns the number of allowed tokens.
long vdamd_json_mask(const void* state, const unsigned char* blob, const long* offsets, long n_tokens,
                     unsigned char* mask) {
  const State base = *static_cast<const State*>(state);
  long allowed = 0;
  for (long i = 0; i < n_tokens; ++i) {
    const long b = offsets[i], e = offsets[i + 1];
    bool ok = e > b;
    if (ok) {
      State s = base;
      ok = feed(s, blob + b, e - b);
    }
    mask[i] = ok ? 1 : 0;
    allowed += ok;
  }
  return allowed;
}

}  // extern "C"
