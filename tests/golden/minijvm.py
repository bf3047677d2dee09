"""minijvm.py — a small interpreter for JVM bytecode, enough to RUN the reference's own compiled Java methods.

TEST INFRASTRUCTURE ONLY (used by tests/golden/make_java_fixtures_jvm.py in the build container, where the
reference's released jar /root/reference/Release/JavaGUI/JTempestSDR.jar can be read; never by the product, never on
the GPU box).  The image has no JVM, so the two rows of SURVEY 8(f) whose reference is Java — plot decimation
(PlotVisualizer.populateData, ZoomableXScale) and mode detection (Main.onIncommingPlot, VideoMode) — could only be
checked against hand transliterations.  This interpreter executes the classes of the jar themselves: class-file
parsing (JVMS 4), a frame / operand-stack machine for the ~130 opcodes javac emits for such code (JVMS 6), Java's
integer and floating-point semantics (32/64-bit wrap-around, truncating division, saturating d2i / d2l, IEEE doubles),
objects as field dictionaries, static initialisers, enums.  What it does NOT interpret is the JDK: the handful of
library classes the methods touch (Math, Object, Enum, Integer / Long / Double boxing, HashMap, String / StringBuilder /
PrintStream) are implemented natively below, and every other class that is not in the jar (Swing widgets and the like)
is an opaque stub whose methods do nothing and return zero / null — the GUI side effects of the methods are irrelevant
to the numbers they compute.
"""
import math
import struct
import zipfile

I32 = 0xFFFFFFFF
I64 = 0xFFFFFFFFFFFFFFFF


def i32(x):
    x &= I32
    return x - (1 << 32) if x & 0x80000000 else x


def i64(x):
    x &= I64
    return x - (1 << 64) if x & (1 << 63) else x


def d2i(x, bits=32):
    if x != x:
        return 0
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    if x >= hi:
        return hi
    if x <= lo:
        return lo
    return int(x)


class JObj:
    def __init__(self, cls):
        self.cls = cls
        self.fields = {}
        self.native = None  # payload of natively implemented classes (boxed value, dict, ...)

    def __repr__(self):
        return f"<{self.cls} {self.native if self.native is not None else ''}>"


class JArr:
    def __init__(self, kind, data):
        self.kind = kind
        self.data = data


class ClassFile:
    def __init__(self, raw):
        self.raw = raw
        self.pos = 0
        assert self.u4() == 0xCAFEBABE
        self.u2(), self.u2()
        n = self.u2()
        self.cp = [None] * n
        i = 1
        while i < n:
            tag = self.u1()
            if tag == 1:
                ln = self.u2()
                self.cp[i] = ("utf8", self.raw[self.pos:self.pos + ln].decode("utf-8", "replace"))
                self.pos += ln
            elif tag == 3:
                self.cp[i] = ("int", i32(self.u4()))
            elif tag == 4:
                self.cp[i] = ("float", struct.unpack(">f", self.take(4))[0])
            elif tag == 5:
                self.cp[i] = ("long", i64(self.u8()))
                i += 1
            elif tag == 6:
                self.cp[i] = ("double", struct.unpack(">d", self.take(8))[0])
                i += 1
            elif tag == 7:
                self.cp[i] = ("class", self.u2())
            elif tag == 8:
                self.cp[i] = ("string", self.u2())
            elif tag in (9, 10, 11):
                self.cp[i] = ({9: "field", 10: "method", 11: "imethod"}[tag], self.u2(), self.u2())
            elif tag == 12:
                self.cp[i] = ("nat", self.u2(), self.u2())
            elif tag == 15:
                self.cp[i] = ("mh", self.u1(), self.u2())
            elif tag == 16:
                self.cp[i] = ("mt", self.u2())
            elif tag == 18:
                self.cp[i] = ("indy", self.u2(), self.u2())
            else:
                raise ValueError(f"constant pool tag {tag}")
            i += 1
        self.access = self.u2()
        self.name = self.cls_name(self.u2())
        sup = self.u2()
        self.super = self.cls_name(sup) if sup else None
        self.interfaces = [self.cls_name(self.u2()) for _ in range(self.u2())]
        self.fields = {}
        for _ in range(self.u2()):
            acc, name, desc = self.u2(), self.utf(self.u2()), self.utf(self.u2())
            const = None
            for _a in range(self.u2()):
                an, ln = self.utf(self.u2()), self.u4()
                body = self.take(ln)
                if an == "ConstantValue":
                    const = struct.unpack(">H", body)[0]
            self.fields[name] = (acc, desc, const)
        self.methods = {}
        for _ in range(self.u2()):
            acc, name, desc = self.u2(), self.utf(self.u2()), self.utf(self.u2())
            code = None
            for _a in range(self.u2()):
                an, ln = self.utf(self.u2()), self.u4()
                body = self.take(ln)
                if an == "Code":
                    max_stack, max_locals, clen = struct.unpack(">HHI", body[:8])
                    code = (max_locals, body[8:8 + clen])
            self.methods[(name, desc)] = (acc, code)

    def take(self, n):
        b = self.raw[self.pos:self.pos + n]
        self.pos += n
        return b

    def u1(self):
        return self.take(1)[0]

    def u2(self):
        return struct.unpack(">H", self.take(2))[0]

    def u4(self):
        return struct.unpack(">I", self.take(4))[0]

    def u8(self):
        return struct.unpack(">Q", self.take(8))[0]

    def utf(self, i):
        return self.cp[i][1]

    def cls_name(self, i):
        return self.utf(self.cp[i][1])

    def member(self, i):
        _, c, nt = self.cp[i]
        return self.cls_name(c), self.utf(self.cp[nt][1]), self.utf(self.cp[nt][2])


def parse_desc(desc):
    """(args kinds, return kind) of a method descriptor; kinds: I J F D A V (Z B C S are I)."""
    args, i = [], 1
    while desc[i] != ")":
        c = desc[i]
        if c in "ZBCSI":
            args.append("I")
        elif c in "JFD":
            args.append(c)
        elif c == "L":
            args.append("A")
            i = desc.index(";", i)
        elif c == "[":
            while desc[i] == "[":
                i += 1
            if desc[i] == "L":
                i = desc.index(";", i)
            args.append("A")
        i += 1
    r = desc[i + 1]
    ret = "I" if r in "ZBCSI" else ("A" if r in "L[" else r)
    return args, ret


DEFAULTS = {"I": 0, "J": 0, "F": 0.0, "D": 0.0, "A": None}


def default_of(desc):
    c = desc[0]
    return 0 if c in "ZBCSIJ" else (0.0 if c in "FD" else None)


class Box:
    """java.lang.Integer / Long / Double as far as boxing, unboxing and HashMap keys go"""
    KIND = {"java/lang/Integer": "I", "java/lang/Long": "J", "java/lang/Double": "D", "java/lang/Boolean": "I"}


class VM:
    def __init__(self, jar_path):
        self.zip = zipfile.ZipFile(jar_path)
        self.names = set(n[:-6] for n in self.zip.namelist() if n.endswith(".class"))
        self.classes = {}
        self.statics = {}
        self.initialised = set()
        self.hooks = {}   # (class, name, desc) -> python callable(vm, args) overriding a method (test stubs)
        self.trace = []   # calls into stubbed / opaque methods, for the curious

    # ---- classes ---------------------------------------------------------------------------------------
    def in_jar(self, name):
        return name in self.names

    def load(self, name):
        if name not in self.classes:
            self.classes[name] = ClassFile(self.zip.read(name + ".class"))
        return self.classes[name]

    def init(self, name):
        if name in self.initialised or not self.in_jar(name):
            return
        self.initialised.add(name)
        cf = self.load(name)
        if cf.super:
            self.init(cf.super)
        st = self.statics.setdefault(name, {})
        for fname, (acc, desc, const) in cf.fields.items():
            if acc & 0x0008:
                st[fname] = default_of(desc) if const is None else self.constant(cf, const)
        if ("<clinit>", "()V") in cf.methods:
            self.invoke(name, "<clinit>", "()V", [])

    def constant(self, cf, i):
        e = cf.cp[i]
        if e[0] in ("int", "float", "long", "double"):
            return e[1]
        if e[0] == "string":
            return cf.utf(e[1])
        if e[0] == "class":
            return JObj("java/lang/Class")
        raise ValueError(e)

    def new(self, name):
        self.init(name)
        o = JObj(name)
        c = name
        while c and self.in_jar(c):
            cf = self.load(c)
            for fname, (acc, desc, _) in cf.fields.items():
                if not acc & 0x0008:
                    o.fields[fname] = default_of(desc)
            c = cf.super
        return o

    def find_method(self, cls, name, desc):
        c = cls
        while c and self.in_jar(c):
            cf = self.load(c)
            if (name, desc) in cf.methods:
                return c, cf.methods[(name, desc)]
            c = cf.super
        return None, None

    def is_instance(self, obj, target):
        if obj is None:
            return False
        if isinstance(obj, (JArr, str)):
            return True
        c = obj.cls
        while c:
            if c == target:
                return True
            if not self.in_jar(c):
                return True  # an opaque superclass chain: give the benefit of the doubt
            cf = self.load(c)
            if target in cf.interfaces:
                return True
            c = cf.super
        return False

    # ---- natives ---------------------------------------------------------------------------------------
    def native(self, cls, name, desc, args):
        """JDK classes the interpreted methods touch.  Returns (handled, value)."""
        if name == "clone" and args and isinstance(args[0], JArr):
            return True, JArr(args[0].kind, list(args[0].data))
        if cls == "java/lang/System" and name == "arraycopy":
            src, sp, dst, dp, n = args
            dst.data[dp:dp + n] = src.data[sp:sp + n]
            return True, None
        if cls == "java/lang/Math":
            f = {"min": min, "max": max, "abs": abs, "floor": math.floor, "ceil": math.ceil, "sqrt": math.sqrt,
                 "log10": lambda x: math.log10(x) if x > 0 else (float("-inf") if x == 0 else float("nan")),
                 "log": lambda x: math.log(x) if x > 0 else (float("-inf") if x == 0 else float("nan")),
                 "pow": math.pow, "exp": math.exp, "sin": math.sin, "cos": math.cos}
            if name == "round":
                v = math.floor(args[0] + 0.5) if args[0] == args[0] else 0.0
                return True, d2i(v, 64 if desc == "(D)J" else 32)
            if name in ("floor", "ceil"):
                return True, float(f[name](args[0]))
            if name in f:
                return True, f[name](*args)
        if cls in Box.KIND:
            if name == "valueOf" and len(args) == 1 and not isinstance(args[0], str):
                o = JObj(cls)
                o.native = args[0]
                return True, o
            if name == "<init>":
                args[0].native = args[1]
                return True, None
            if name in ("intValue", "longValue", "doubleValue", "booleanValue"):
                v = args[0].native
                return True, (float(v) if name == "doubleValue" else (d2i(v, 64 if name == "longValue" else 32) if isinstance(v, float) else v))
            if name == "equals":
                return True, int(isinstance(args[1], JObj) and args[1].cls == cls and args[1].native == args[0].native)
            if name == "hashCode":
                return True, i32(hash(args[0].native))
        if cls in ("java/util/HashMap", "java/util/Map", "java/util/Hashtable"):
            if name == "<init>":
                args[0].native = {}
                return True, None
            key = lambda k: (k.cls, k.native) if isinstance(k, JObj) and k.native is not None else k  # noqa: E731
            if name == "get":
                return True, args[0].native.get(key(args[1]))
            if name == "put":
                old = args[0].native.get(key(args[1]))
                args[0].native[key(args[1])] = args[2]
                return True, old
            if name == "containsKey":
                return True, int(key(args[1]) in args[0].native)
            if name == "clear":
                args[0].native.clear()
                return True, None
            if name == "size":
                return True, len(args[0].native)
        if cls == "java/lang/Enum":
            if name == "<init>":
                args[0].fields["$name"], args[0].fields["$ordinal"] = args[1], args[2]
                return True, None
            if name == "ordinal":
                return True, args[0].fields["$ordinal"]
            if name == "name" or name == "toString":
                return True, args[0].fields["$name"]
        if cls == "java/lang/Object":
            if name == "<init>":
                return True, None
            if name == "clone":
                a = args[0]
                return True, JArr(a.kind, list(a.data)) if isinstance(a, JArr) else a
            if name == "hashCode":
                return True, i32(id(args[0]))
            if name == "equals":
                return True, int(args[0] is args[1])
        if cls == "java/lang/String" and name == "format":
            return True, "<formatted>"
        if cls == "java/lang/StringBuilder":
            if name == "<init>":
                args[0].native = ""
                return True, None
            if name == "append":
                args[0].native += str(args[1])
                return True, args[0]
            if name == "toString":
                return True, args[0].native
        return False, None

    # ---- invocation ------------------------------------------------------------------------------------
    def invoke(self, cls, name, desc, args, virtual=False):
        if (cls, name, desc) in self.hooks:
            return self.hooks[(cls, name, desc)](self, args)
        recv_cls = cls
        if virtual and args and isinstance(args[0], JObj):
            recv_cls = args[0].cls
            if (recv_cls, name, desc) in self.hooks:
                return self.hooks[(recv_cls, name, desc)](self, args)
        owner, m = self.find_method(recv_cls if self.in_jar(recv_cls) else cls, name, desc)
        if m is None:
            # not in the jar: a JDK class implemented natively, or an opaque stub (Swing and friends)
            for c in (recv_cls, cls):
                ok, v = self.native(c, name, desc, args)
                if ok:
                    return v
            # walk the jar part of the receiver's hierarchy up to its first foreign superclass: natives may live there
            c = recv_cls
            while c and self.in_jar(c):
                c = self.load(c).super
            if c:
                ok, v = self.native(c, name, desc, args)
                if ok:
                    return v
            self.trace.append((cls, name, desc))
            return DEFAULTS.get(parse_desc(desc)[1])
        acc, code = m
        if code is None:  # native / abstract in the jar
            self.trace.append((owner, name, desc))
            return DEFAULTS.get(parse_desc(desc)[1])
        if acc & 0x0008:
            self.init(owner)
        return self.run(self.load(owner), code, args, desc, static=bool(acc & 0x0008))

    def run(self, cf, code, args, desc, static):
        max_locals, bc = code
        loc = [None] * (max_locals + 2)
        kinds = parse_desc(desc)[0]
        slot = 0
        ai = 0
        if not static:
            loc[0] = args[0]
            slot, ai = 1, 1
        for k in kinds:
            loc[slot] = args[ai]
            slot += 2 if k in "JD" else 1
            ai += 1
        st = []   # operand stack of (kind, value); kind in I J F D A
        pc = 0
        push = st.append

        def s2(off):
            return struct.unpack(">h", bc[off:off + 2])[0]

        def u2(off):
            return struct.unpack(">H", bc[off:off + 2])[0]

        def s4(off):
            return struct.unpack(">i", bc[off:off + 4])[0]

        while True:
            op = bc[pc]
            # ---- constants
            if op == 0x00:
                pc += 1
            elif op == 0x01:
                push(("A", None)); pc += 1
            elif 0x02 <= op <= 0x08:
                push(("I", op - 3)); pc += 1
            elif op in (0x09, 0x0A):
                push(("J", op - 9)); pc += 1
            elif 0x0B <= op <= 0x0D:
                push(("F", float(op - 0x0B))); pc += 1
            elif op in (0x0E, 0x0F):
                push(("D", float(op - 0x0E))); pc += 1
            elif op == 0x10:
                push(("I", struct.unpack(">b", bc[pc + 1:pc + 2])[0])); pc += 2
            elif op == 0x11:
                push(("I", s2(pc + 1))); pc += 3
            elif op in (0x12, 0x13, 0x14):
                idx = bc[pc + 1] if op == 0x12 else u2(pc + 1)
                e = cf.cp[idx]
                v = self.constant(cf, idx)
                push(({"int": "I", "float": "F", "long": "J", "double": "D"}.get(e[0], "A"), v))
                pc += 2 if op == 0x12 else 3
            # ---- loads / stores
            elif 0x15 <= op <= 0x19:
                push(("IJFDA"[op - 0x15], loc[bc[pc + 1]])); pc += 2
            elif 0x1A <= op <= 0x2D:
                k = (op - 0x1A) // 4
                push(("IJFDA"[k], loc[(op - 0x1A) % 4])); pc += 1
            elif 0x2E <= op <= 0x35:  # array loads
                i = st.pop()[1]; a = st.pop()[1]
                v = a.data[i]
                push(({0x2E: "I", 0x2F: "J", 0x30: "F", 0x31: "D", 0x32: "A", 0x33: "I", 0x34: "I", 0x35: "I"}[op], v)); pc += 1
            elif 0x36 <= op <= 0x3A:
                loc[bc[pc + 1]] = st.pop()[1]; pc += 2
            elif 0x3B <= op <= 0x4E:
                loc[(op - 0x3B) % 4] = st.pop()[1]; pc += 1
            elif 0x4F <= op <= 0x56:  # array stores
                v = st.pop()[1]; i = st.pop()[1]; a = st.pop()[1]
                if op == 0x54:
                    v = struct.unpack("b", struct.pack("B", v & 0xFF))[0]
                a.data[i] = v; pc += 1
            # ---- stack
            elif op == 0x57:
                st.pop(); pc += 1
            elif op == 0x58:
                if st.pop()[0] not in "JD":
                    st.pop()
                pc += 1
            elif op == 0x59:
                push(st[-1]); pc += 1
            elif op == 0x5A:  # dup_x1
                a = st.pop(); b = st.pop(); st.extend([a, b, a]); pc += 1
            elif op == 0x5B:  # dup_x2
                a = st.pop(); b = st.pop()
                if b[0] in "JD":
                    st.extend([a, b, a])
                else:
                    c = st.pop(); st.extend([a, c, b, a])
                pc += 1
            elif op == 0x5C:  # dup2
                if st[-1][0] in "JD":
                    push(st[-1])
                else:
                    st.extend([st[-2], st[-1]])
                pc += 1
            elif op == 0x5D:  # dup2_x1
                a = st.pop()
                if a[0] in "JD":
                    b = st.pop(); st.extend([a, b, a])
                else:
                    b = st.pop(); c = st.pop(); st.extend([b, a, c, b, a])
                pc += 1
            elif op == 0x5F:
                a = st.pop(); b = st.pop(); st.extend([a, b]); pc += 1
            # ---- arithmetic
            elif 0x60 <= op <= 0x77:
                k = "IJFD"[(op - 0x60) % 4]
                grp = (op - 0x60) // 4
                if grp == 5:  # neg
                    a = st.pop()[1]
                    r = -a
                else:
                    b = st.pop()[1]; a = st.pop()[1]
                    if grp == 0:
                        r = a + b
                    elif grp == 1:
                        r = a - b
                    elif grp == 2:
                        r = a * b
                    elif grp == 3:
                        if k in "IJ":
                            if b == 0:
                                raise ZeroDivisionError("java.lang.ArithmeticException")
                            r = abs(a) // abs(b) * (1 if (a < 0) == (b < 0) else -1)
                        else:
                            r = (a / b) if b != 0 else (float("nan") if a == 0 or a != a else math.copysign(float("inf"), a) * math.copysign(1.0, b))
                    else:  # rem
                        if k in "IJ":
                            r = abs(a) % abs(b) * (1 if a >= 0 else -1)
                        else:
                            r = math.fmod(a, b) if b != 0 else float("nan")
                if k == "I":
                    r = i32(r)
                elif k == "J":
                    r = i64(r)
                elif k == "F":
                    r = struct.unpack("f", struct.pack("f", r))[0]
                push((k, r)); pc += 1
            elif 0x78 <= op <= 0x83:  # shifts and bitwise, int / long alternating
                k = "IJ"[(op - 0x78) % 2]
                b = st.pop()[1]; a = st.pop()[1]
                bits = 32 if k == "I" else 64
                grp = (op - 0x78) // 2
                if grp == 0:
                    r = a << (b & (bits - 1))
                elif grp == 1:
                    r = a >> (b & (bits - 1))
                elif grp == 2:
                    r = (a & ((1 << bits) - 1)) >> (b & (bits - 1))
                elif grp == 3:
                    r = a & b
                elif grp == 4:
                    r = a | b
                else:
                    r = a ^ b
                push((k, i32(r) if k == "I" else i64(r))); pc += 1
            elif op == 0x84:
                idx = bc[pc + 1]
                loc[idx] = i32(loc[idx] + struct.unpack(">b", bc[pc + 2:pc + 3])[0]); pc += 3
            # ---- conversions
            elif 0x85 <= op <= 0x93:
                v = st.pop()[1]
                t = {0x85: "J", 0x86: "F", 0x87: "D", 0x88: "I", 0x89: "F", 0x8A: "D", 0x8B: "I", 0x8C: "J", 0x8D: "D",
                     0x8E: "I", 0x8F: "J", 0x90: "F", 0x91: "I", 0x92: "I", 0x93: "I"}[op]
                if op in (0x8B, 0x8E):
                    v = d2i(v, 32)
                elif op in (0x8C, 0x8F):
                    v = d2i(v, 64)
                elif op == 0x88:
                    v = i32(v)
                elif op == 0x91:
                    v = struct.unpack("b", struct.pack("B", v & 0xFF))[0]
                elif op == 0x92:
                    v &= 0xFFFF
                elif op == 0x93:
                    v = struct.unpack("h", struct.pack("H", v & 0xFFFF))[0]
                elif t == "F":
                    v = struct.unpack("f", struct.pack("f", float(v)))[0]
                elif t == "D":
                    v = float(v)
                push((t, v)); pc += 1
            # ---- comparisons
            elif op == 0x94:
                b = st.pop()[1]; a = st.pop()[1]
                push(("I", (a > b) - (a < b))); pc += 1
            elif 0x95 <= op <= 0x98:
                b = st.pop()[1]; a = st.pop()[1]
                if a != a or b != b:
                    r = 1 if op in (0x96, 0x98) else -1
                else:
                    r = (a > b) - (a < b)
                push(("I", r)); pc += 1
            elif 0x99 <= op <= 0x9E:
                a = st.pop()[1]
                t = [a == 0, a != 0, a < 0, a >= 0, a > 0, a <= 0][op - 0x99]
                pc += s2(pc + 1) if t else 3
            elif 0x9F <= op <= 0xA4:
                b = st.pop()[1]; a = st.pop()[1]
                t = [a == b, a != b, a < b, a >= b, a > b, a <= b][op - 0x9F]
                pc += s2(pc + 1) if t else 3
            elif op in (0xA5, 0xA6):
                b = st.pop()[1]; a = st.pop()[1]
                t = (a is b) if op == 0xA5 else (a is not b)
                pc += s2(pc + 1) if t else 3
            elif op == 0xA7:
                pc += s2(pc + 1)
            elif op == 0xAA:  # tableswitch
                base = (pc + 4) & ~3
                dflt, lo, hi = s4(base), s4(base + 4), s4(base + 8)
                v = st.pop()[1]
                pc += s4(base + 12 + 4 * (v - lo)) if lo <= v <= hi else dflt
            elif op == 0xAB:  # lookupswitch
                base = (pc + 4) & ~3
                dflt, n = s4(base), s4(base + 4)
                v = st.pop()[1]
                off = dflt
                for j in range(n):
                    if s4(base + 8 + 8 * j) == v:
                        off = s4(base + 12 + 8 * j)
                        break
                pc += off
            elif 0xAC <= op <= 0xB0:
                return st.pop()[1]
            elif op == 0xB1:
                return None
            # ---- fields
            elif op in (0xB2, 0xB3):
                c, n, d = cf.member(u2(pc + 1))
                owner = c
                while owner and self.in_jar(owner) and n not in self.load(owner).fields:
                    owner = self.load(owner).super
                if owner and self.in_jar(owner):
                    self.init(owner)
                    if op == 0xB2:
                        push((parse_kind(d), self.statics[owner][n]))
                    else:
                        self.statics[owner][n] = st.pop()[1]
                else:  # a JDK static (System.out, ...): opaque
                    if op == 0xB2:
                        push((parse_kind(d), JObj(d[1:-1]) if d[0] == "L" else default_of(d)))
                    else:
                        st.pop()
                pc += 3
            elif op == 0xB4:
                c, n, d = cf.member(u2(pc + 1))
                o = st.pop()[1]
                push((parse_kind(d), o.fields.get(n, default_of(d)))); pc += 3
            elif op == 0xB5:
                c, n, d = cf.member(u2(pc + 1))
                v = st.pop()[1]; o = st.pop()[1]
                o.fields[n] = v; pc += 3
            # ---- invocations
            elif 0xB6 <= op <= 0xB9:
                c, n, d = cf.member(u2(pc + 1))
                kinds_, ret = parse_desc(d)
                nargs = len(kinds_) + (0 if op == 0xB8 else 1)
                a = [x[1] for x in st[len(st) - nargs:]] if nargs else []
                del st[len(st) - nargs:]
                if op == 0xB8:
                    self.init(c)
                if op != 0xB8 and a[0] is None:
                    raise RuntimeError(f"NullPointerException calling {c}.{n}{d}")
                v = self.invoke(c, n, d, a, virtual=op in (0xB6, 0xB9))
                if ret != "V":
                    push((ret, v))
                pc += 5 if op == 0xB9 else 3
            elif op == 0xBB:
                push(("A", self.new(cf.cls_name(u2(pc + 1))))); pc += 3
            elif op == 0xBC:
                n = st.pop()[1]
                t = bc[pc + 1]
                push(("A", JArr(t, [0.0 if t in (6, 7) else 0] * n))); pc += 2
            elif op == 0xBD:
                n = st.pop()[1]
                push(("A", JArr("A", [None] * n))); pc += 3
            elif op == 0xBE:
                push(("I", len(st.pop()[1].data))); pc += 1
            elif op == 0xBF:
                raise RuntimeError(f"athrow: {st.pop()[1]}")
            elif op == 0xC0:
                pc += 3  # checkcast: trusted
            elif op == 0xC1:
                o = st.pop()[1]
                push(("I", int(self.is_instance(o, cf.cls_name(u2(pc + 1)))))); pc += 3
            elif op in (0xC2, 0xC3):
                st.pop(); pc += 1
            elif op in (0xC6, 0xC7):
                a = st.pop()[1]
                t = (a is None) if op == 0xC6 else (a is not None)
                pc += s2(pc + 1) if t else 3
            elif op == 0xC8:
                pc += s4(pc + 1)
            else:
                raise NotImplementedError(f"opcode 0x{op:02x} in {cf.name}")


def parse_kind(desc):
    c = desc[0]
    return "I" if c in "ZBCSI" else ("A" if c in "L[" else c)
