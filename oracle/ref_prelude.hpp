// oracle/ref_prelude.hpp -- the ONLY hand-written code inside oracle/_ref (test infrastructure).
// Everything else in libk4ref.so is the reference's own engine source, respelled by oracle/make_ref.py.
// Each member below stands in for a reference member whose body is a call into the .NET runtime
// (list = make_ref.py EXCLUDED; paths relative to /root/reference/src/K4os.Compression.LZ4/).
#pragma once
#include <alloca.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace k4ref {
// C# built-in integer names (ECMA-334 8.3): fixed widths, `long` is 64-bit as in LP64
typedef uint8_t byte;
typedef int8_t sbyte;
typedef uint16_t ushort;
typedef uint32_t uint;
typedef uint64_t ulong;

#ifndef K4REF_ASSERT
#define K4REF_ASSERT(v) ((void) (v))          // [Conditional("DEBUG")]: a release build drops the call
#endif

// C# lets an enum be compared with the literal 0 (ECMA-334 10.2.4); `enum class` needs it spelled out
template <class E> constexpr typename std::enable_if<std::is_enum<E>::value, bool>::type operator!=(E a, int b) { return (int) a != b; }
template <class E> constexpr typename std::enable_if<std::is_enum<E>::value, bool>::type operator==(E a, int b) { return (int) a == b; }

// R16: try { A } finally { B }
template <class F> struct K4RefFinally { F f; K4RefFinally(F g) : f(g) {} ~K4RefFinally() { f(); } };

// Internal/Mem.cs:47-52,:69-71,:79-81,:103-106,:112-114,:144-145,:151-162
#define K4REF_MEMBERS_Mem \
	static constexpr bool System32 = sizeof(void*) < sizeof(ulong);                                          /* :50-52 */ \
	static void CpBlk(void* target, const void* source, uint length) { memcpy(target, source, length); }   /* :69-71 Unsafe.CopyBlockUnaligned */ \
	static void ZBlk(void* target, byte value, uint length) { memset(target, value, length); }             /* :79-81 Unsafe.InitBlockUnaligned */ \
	static void Move(byte* target, byte* source, int length) { memmove(target, source, (size_t) length); } /* :103-106 Buffer.MemoryCopy */ \
	static void* Alloc(int size) { return malloc((size_t) size); }                                         /* :112-114 Marshal.AllocHGlobal */ \
	static void Free(void* ptr) { free(ptr); }                                                             /* :144-145 Marshal.FreeHGlobal */ \
	template <class T, size_t N> static T* CloneArray(T (&array)[N]) {                                     /* :151-162 */ \
		T* target = (T*) Alloc((int) sizeof(array)); memcpy(target, array, sizeof(array)); return target; }

// Engine/LL.tools.cs:21-27 (Assert; Enforce32 / Algorithm, :29-36, are the managed dispatch the C entry points replace)
#define K4REF_MEMBERS_LL \
	static void Assert(bool value) { K4REF_ASSERT(value); }

// Internal/PinnedMemory.cs:63-77 (Alloc), :31-33 (Reference<T>), :121-134 (Free): pooled or native, not zeroed when zero == false
struct PinnedMemory {
	byte* _pointer = nullptr;
	static void Alloc(PinnedMemory& memory, int size, bool zero = true) { memory._pointer = (byte*) (zero ? calloc(1, (size_t) size) : malloc((size_t) size)); }
	template <class T> T* Reference() { return (T*) _pointer; }
	void Free() { free(_pointer); _pointer = nullptr; }
};

// System.Numerics.BitOperations.TrailingZeroCount(ulong) (used by LL64.tools.cs:57-59 under NET5_0_OR_GREATER)
struct BitOperations {
	static int TrailingZeroCount(ulong v) { return v ? __builtin_ctzll(v) : 64; }
};

// ---- the LZ4Pickler header helpers (make_ref.py ONLY["LZ4Pickler"]): the .NET types their bodies touch
// System.Span<T> / ReadOnlySpan<T>: a pointer and a length; indexer, Length, Slice(start) -- all the helpers use.  Out of range
// is an IndexOutOfRangeException / ArgumentOutOfRangeException there, K4RefManaged{3} here.
struct K4RefManaged { int kind; const char* message; };      // kind 1 ArgumentException, 2 InvalidDataException, 3 index/range
template <class T> struct Span {
	T* _p; int Length;
	Span(T* p, int n) : _p(p), Length(n) {}
	template <class U> Span(const Span<U>& o) : _p((T*) o._p), Length(o.Length) {}
	T& operator[](int i) const { if ((uint) i >= (uint) Length) throw K4RefManaged{3, "index"}; return _p[i]; }
	Span Slice(int start) const { if ((uint) start > (uint) Length) throw K4RefManaged{3, "start"}; return Span(_p + start, Length - start); }
};
template <class T> using ReadOnlySpan = Span<const T>;
// System.Diagnostics.Debug.Assert is [Conditional("DEBUG")]; K4REF_CHECKED keeps it so that the pins can see it hold
struct Debug {
	static void Assert(bool v, const char* = nullptr) {
#ifdef K4REF_CHECKED
		if (!v) throw K4RefManaged{4, "Debug.Assert"};
#else
		(void) v;
#endif
	}
};
// LZ4Pickler.unpickle.cs:163-181: a readonly struct of three auto-properties (make_ref.py TOPLEVEL_SUPPLIED); the constructor is :170-178
struct PickleHeader {
	ushort DataOffset; ushort Flags; int ResultLength;
	bool IsCompressed() const { return (Flags & 0x0001) != 0; }                                           /* :168 */
	PickleHeader(ushort dataOffset, int resultLength, bool compressed)
		: DataOffset(dataOffset), Flags((ushort) ((compressed ? 0x0001 : 0x0000) << 0 | 0)), ResultLength(resultLength) {}
};
// LZ4Pickler.pickle.cs:214-219 (PokeN), :230-231 (UnexpectedVersion); LZ4Pickler.unpickle.cs:150-158 (PeekN), :160-161 (CorruptedPickle):
// the same checks, the little-endian Unsafe.CopyBlockUnaligned of `size` bytes spelled memcpy
#define K4REF_MEMBERS_LZ4Pickler \
	static void PokeN(Span<byte> target, int value, int size) {                                           /* pickle.cs:214-219 */ \
		if (size < 0 || size > (int) sizeof(int) || target.Length < size) throw K4RefManaged{1, "Unexpected size"}; \
		if (size > 0) memcpy(&target[0], &value, (size_t) size); } \
	static int PeekN(ReadOnlySpan<byte> bytes, int size) {                                                /* unpickle.cs:150-158 */ \
		int result = 0; \
		if (size < 0 || size > (int) sizeof(int) || size > bytes.Length) throw CorruptedPickle("Unexpected field size"); \
		memcpy(&result, bytes._p, (size_t) size); return result; } \
	static K4RefManaged UnexpectedVersion(int) { return K4RefManaged{1, "Unexpected pickle version"}; }   /* pickle.cs:230-231 */ \
	static K4RefManaged CorruptedPickle(const char* m) { return K4RefManaged{2, m}; }                     /* unpickle.cs:160-161 */
} // namespace k4ref
