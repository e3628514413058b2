"""goref.py — a second, independent CPU restatement of the reference's packet decode path, in pure Python.

TEST INFRASTRUCTURE ONLY. Written in round 2 straight from the Go source (without consulting oracle/alac_oracle.c),
so that the C oracle, this file, the hand-derived KATs and the HIP kernels are four readings of the same algorithm
that must agree byte for byte. It follows the reference's control flow statement by statement (block passes:
DynDecomp over the block, then UnpcBlock, then Write*), which is NOT how the product is structured.

Go semantics are made explicit:
  * int32 / uint32 / uint16 / int16 / uint8 arithmetic wraps (helpers i32, u32, ...);
  * `x << n` with n >= width is 0, unsigned `x >> n` with n >= width is 0, signed `>>` sign-fills;
  * slice expressions and index expressions that would panic in Go raise GoPanic (the reference has no recover();
    the product reports ALACGPU_ERR_MALFORMED for these inputs);
  * a fresh PacketDecoder per packet (cap(Buf) == len(Buf), zeroed scratch), as DecodePacket on a new decoder.

Reference lines (paths relative to the reference tree) are cited per function.
"""

M32 = 0xFFFFFFFF


class GoPanic(Exception):
    """The Go code would panic here (slice bounds / index out of range)."""


class DecodeError(Exception):
    """ErrDecode wrapping an internal sentinel; .chain is the list of wrapped context strings + sentinel."""

    def __init__(self, *chain):
        super().__init__(": ".join(chain))
        self.chain = chain


ErrBitstreamOverrun = "alac: bitstream overrun"
ErrSampleOverrun = "alac: sample count exceeds buffer"
ErrInvalidHeader = "alac: invalid frame header"
ErrInvalidShift = "alac: invalid bytesShifted value"
ErrUnsupportedElement = "alac: unsupported element type (CCE/PCE)"


def u32(x):
    return x & M32


def i32(x):
    x &= M32
    return x - (1 << 32) if x & 0x80000000 else x


def i16(x):
    x &= 0xFFFF
    return x - (1 << 16) if x & 0x8000 else x


def i8(x):
    x &= 0xFF
    return x - 256 if x & 0x80 else x


def shl32(x, n):
    """Go: uint32(x) << n (n unsigned); 0 for n >= 32."""
    return 0 if n >= 32 else (x << n) & M32


def shr32(x, n):
    """Go: uint32(x) >> n; 0 for n >= 32."""
    return 0 if n >= 32 else (x & M32) >> n


def sar32(x, n):
    """Go: int32(x) >> n; sign fill for n >= 32."""
    x = i32(x)
    return (-1 if x < 0 else 0) if n >= 32 else x >> n


def shl_i32(x, n):
    """Go: int32(x) << n, wrapping."""
    return i32(shl32(u32(x), n))


def leading_zeros32(x):
    x &= M32
    return 32 - x.bit_length()


# ---- internal/alac/bitbuffer.go:25-123 ---------------------------------------------------------------------------
class BitBuffer:
    def __init__(self):
        self.Buf = b""
        self.Pos = 0
        self.BitIdx = 0
        self.Size = 0

    def Reset(self, data):  # bitbuffer.go:36-51, fresh storage: cap == len == len(data) + 4
        self.Buf = bytes(data) + b"\0\0\0\0"
        self.Pos = 0
        self.BitIdx = 0
        self.Size = len(data)

    def _window(self, n):  # b.Buf[b.Pos : b.Pos+n : b.Pos+n]
        if self.Pos < 0 or self.Pos + n > len(self.Buf):
            raise GoPanic("slice bounds out of range [%d:%d] with capacity %d" % (self.Pos, self.Pos + n, len(self.Buf)))
        return self.Buf[self.Pos:self.Pos + n]

    def Read(self, numBits):  # bitbuffer.go:55-69
        w = self._window(3)
        returnBits = (w[0] << 16) | (w[1] << 8) | w[2]
        returnBits = shl32(returnBits, self.BitIdx) & 0x00FFFFFF
        returnBits = shr32(returnBits, u32(24 - numBits))
        self.BitIdx = u32(self.BitIdx + numBits)
        self.Pos += self.BitIdx >> 3
        self.BitIdx &= 7
        return returnBits

    def ReadSmall(self, numBits):  # bitbuffer.go:73-85 (uint16 arithmetic)
        w = self._window(2)
        returnBits = (w[0] << 8) | w[1]
        returnBits = (returnBits << self.BitIdx) & 0xFFFF if self.BitIdx < 16 else 0
        sh = (16 - numBits) & 0xFFFF
        returnBits = returnBits >> sh if sh < 16 else 0
        self.BitIdx = u32(self.BitIdx + numBits)
        self.Pos += self.BitIdx >> 3
        self.BitIdx &= 7
        return returnBits & 0xFF

    def ReadOne(self):  # bitbuffer.go:88-96
        if self.Pos < 0 or self.Pos >= len(self.Buf):
            raise GoPanic("index out of range")
        returnBit = (self.Buf[self.Pos] >> (7 - self.BitIdx)) & 1
        self.BitIdx += 1
        self.Pos += self.BitIdx >> 3
        self.BitIdx &= 7
        return returnBit

    def Advance(self, numBits):  # bitbuffer.go:99-103 (BitIdx is uint32 and wraps)
        self.BitIdx = u32(self.BitIdx + numBits)
        self.Pos += self.BitIdx >> 3
        self.BitIdx &= 7

    def ByteAlign(self):  # bitbuffer.go:106-112
        if self.BitIdx == 0:
            return
        self.Advance(8 - self.BitIdx)

    def PastEnd(self):  # bitbuffer.go:115-117
        return self.Pos >= self.Size

    def Copy(self):  # bitbuffer.go:121-123
        c = BitBuffer()
        c.Buf, c.Pos, c.BitIdx, c.Size = self.Buf, self.Pos, self.BitIdx, self.Size
        return c


# ---- internal/alac/golomb.go ----------------------------------------------------------------------------------------
QB_SHIFT = 9
QUANT_BITS = 1 << QB_SHIFT
MMUL_SHIFT = 2
MDEN_SHIFT = QB_SHIFT - MMUL_SHIFT - 1
MOFF = 1 << (MDEN_SHIFT - 2)
BITOFF = 24
MAX_PREFIX_16 = 9
MAX_PREFIX_32 = 9
MAX_DATATYPE_16 = 16
N_MAX_MEAN_CLAMP = 0xFFFF
N_MEAN_CLAMP_VAL = 0xFFFF
MAX_ZERO_RUN = 65535


class AGParams:  # golomb.go:44-65
    def __init__(self, meanBase, partBound, kBase, frameWin, sampleWin, maxrun):
        self.MB = self.MB0 = u32(meanBase)
        self.PB = u32(partBound)
        self.KB = u32(kBase)
        self.WB = u32(shl32(1, self.KB) - 1)
        self.QB = u32(QUANT_BITS - partBound)
        self.FW, self.SW, self.MaxRun = frameWin, sampleWin, maxrun


def lead(m):  # golomb.go:69-71
    return leading_zeros32(u32(m))


def lg3a(x):  # golomb.go:74-76
    return 31 - lead(i32(x + 3))


def read32bit(buf, offset):  # golomb.go:81-83: binary.BigEndian.Uint32(buf[offset:])
    if offset < 0 or offset > len(buf):
        raise GoPanic("slice bounds out of range [%d:%d]" % (offset, len(buf)))
    if len(buf) - offset < 4:
        raise GoPanic("index out of range [3] with length %d" % (len(buf) - offset))
    return int.from_bytes(buf[offset:offset + 4], "big")


def getStreamBits(inp, bitOffset, numBits):  # golomb.go:86-108
    byteOffset = bitOffset // 8
    load1 = read32bit(inp, byteOffset)
    if u32(numBits + (bitOffset & 7)) > 32:
        result = shl32(load1, bitOffset & 7)
        if byteOffset + 4 >= len(inp):
            raise GoPanic("index out of range [%d] with length %d" % (byteOffset + 4, len(inp)))
        load2 = inp[byteOffset + 4]
        load2shift = u32(8 - u32(numBits + (bitOffset & 7) - 32))
        load2 = shr32(load2, load2shift)
        result = shr32(result, u32(32 - numBits))
        result |= load2
        return result
    result = shr32(load1, u32(32 - numBits - (bitOffset & 7)))
    if numBits < 32:
        result &= u32(shl32(1, numBits) - 1)
    return result


def dynGet(inp, bitPos, golombM, golombK):  # golomb.go:112-144
    tempBits = bitPos
    streamLong = read32bit(inp, tempBits >> 3)
    streamLong = shl32(streamLong, tempBits & 7)
    pre = lead(i32(~streamLong))
    if pre >= MAX_PREFIX_16:
        pre = MAX_PREFIX_16
        tempBits = u32(tempBits + pre)
        streamLong = shl32(streamLong, pre)
        result = shr32(streamLong, 32 - MAX_DATATYPE_16)
        tempBits = u32(tempBits + MAX_DATATYPE_16)
        return result, tempBits
    tempBits = u32(tempBits + pre + 1)
    streamLong = shl32(streamLong, pre + 1)
    val = shr32(streamLong, u32(32 - golombK))
    tempBits = u32(tempBits + golombK)
    if val < 2:
        result = u32(pre * golombM)
        tempBits = u32(tempBits - 1)
    else:
        result = u32(pre * golombM + val - 1)
    return result, tempBits


def DynDecomp(params, bitBuf, predCoefs, numSamples, maxSize, trace=None):  # golomb.go:148-253
    if bitBuf.Pos < 0 or bitBuf.Pos > len(bitBuf.Buf):
        raise GoPanic("slice bounds out of range [%d:%d]" % (bitBuf.Pos, len(bitBuf.Buf)))
    inp = bitBuf.Buf[bitBuf.Pos:]
    startPos = bitBuf.BitIdx
    maxPos = u32(u32(bitBuf.Size - bitBuf.Pos) * 8)
    bitPos = startPos
    if numSamples < 0 or numSamples > len(predCoefs):  # predCoefs[:numSamples:numSamples]
        raise GoPanic("slice bounds out of range [:%d] with capacity %d" % (numSamples, len(predCoefs)))
    meanAccum = params.MB0
    zmode = 0
    count = 0
    pbLocal, kbLocal, wbLocal = params.PB, params.KB, params.WB
    while count < numSamples:
        if bitPos >= maxPos:
            raise DecodeError(ErrBitstreamOverrun)
        m = meanAccum >> QB_SHIFT
        k = min(lg3a(i32(m)), i32(kbLocal))
        m = u32(shl32(1, u32(k)) - 1)
        streamLong = read32bit(inp, bitPos >> 3)
        streamLong = shl32(streamLong, bitPos & 7)
        residual = lead(i32(~streamLong))
        if residual >= MAX_PREFIX_32:
            residual = getStreamBits(inp, u32(bitPos + MAX_PREFIX_32), u32(maxSize))
            bitPos = u32(bitPos + MAX_PREFIX_32 + u32(maxSize))
        else:
            bitPos = u32(bitPos + residual + 1)
            if k != 1:
                streamLong = shl32(streamLong, residual + 1)
                v = shr32(streamLong, u32(32 - u32(k)))
                if v >= 2:
                    residual = u32(residual * m + v - 1)
                    bitPos = u32(bitPos + u32(k))
                else:
                    residual = u32(residual * m)
                    bitPos = u32(bitPos + u32(k) - 1)
        ndecode = u32(residual + u32(zmode))
        multiplier = i32(-(ndecode & 1))
        multiplier |= 1
        dl = i32(i32(u32(ndecode + 1) >> 1) * multiplier)
        predCoefs[count] = dl
        count += 1
        meanAccum = u32(pbLocal * u32(residual + u32(zmode)) + meanAccum - (u32(pbLocal * meanAccum) >> QB_SHIFT))
        if residual > N_MAX_MEAN_CLAMP:
            meanAccum = N_MEAN_CLAMP_VAL
        if trace is not None:
            trace.append(("code", count - 1, k, residual, zmode, dl, meanAccum, bitPos))
        zmode = 0
        if u32(meanAccum << MMUL_SHIFT) < QUANT_BITS and count < numSamples:
            zmode = 1
            k32 = max(i32(lead(i32(meanAccum)) - BITOFF + i32((u32(meanAccum + MOFF)) >> MDEN_SHIFT)), 0)
            mz = u32(shl32(1, u32(k32)) - 1) & wbLocal
            residual, bitPos = dynGet(inp, bitPos, mz, u32(k32))
            if count + residual > numSamples:
                raise DecodeError(ErrSampleOverrun)
            end = count + residual
            for j in range(count, end):
                predCoefs[j] = 0
            count = end
            if residual >= MAX_ZERO_RUN:
                zmode = 0
            meanAccum = 0
            if trace is not None:
                trace.append(("zrun", k32, mz, residual, bitPos))
    bitsConsumed = u32(bitPos - startPos)
    bitBuf.Advance(bitsConsumed)


# ---- internal/alac/predictor.go -------------------------------------------------------------------------------------
NUM_ACTIVE_DELTA = 31
MAX_COEFS = 32


def signOfInt(val):  # predictor.go:35-39
    negiShift = i32(u32(-val) >> 31)
    return negiShift | sar32(val, 31)


def UnpcBlock(pc1, out, num, coefs, numActive, chanBits, denShift, trace=None):  # predictor.go:45-94
    chanShift = u32(32 - chanBits)
    denHalf = 0
    if denShift > 0:
        denHalf = i32(shl32(1, denShift - 1))
    if len(out) < 1 or len(pc1) < 1:
        raise GoPanic("index out of range [0]")
    out[0] = pc1[0]
    if numActive == 0:
        if num > 1 and pc1 is not out:
            if num > len(out) or num > len(pc1):
                raise GoPanic("slice bounds out of range [:%d]" % num)
            out[1:num] = pc1[1:num]
        return
    if numActive == NUM_ACTIVE_DELTA:
        prev = out[0]
        for idx in range(1, num):
            if idx >= len(pc1) or idx >= len(out):
                raise GoPanic("index out of range [%d]" % idx)
            dl = i32(pc1[idx] + prev)
            prev = sar32(shl_i32(dl, chanShift), chanShift)
            out[idx] = prev
        return
    for idx in range(1, numActive + 1):  # warm-up, predictor.go:76-79
        if idx >= len(pc1) or idx >= len(out):
            raise GoPanic("index out of range [%d] with length %d" % (idx, min(len(pc1), len(out))))
        dl = i32(pc1[idx] + out[idx - 1])
        out[idx] = sar32(shl_i32(dl, chanShift), chanShift)
    if numActive in (4, 5, 6, 8):
        unpcBlockFixed(numActive, pc1, out, num, coefs, chanShift, denShift, denHalf, trace)
    else:
        lim = numActive + 1
        unpcBlockGeneral(pc1, out, num, coefs, numActive, lim, chanShift, denShift, denHalf, trace)


def unpcBlockFixed(order, pc1, out, num, coefs, chanShift, denShift, denHalf, trace=None):
    """unpcBlock4 / 5 / 6 / 8 (predictor.go:99-193, 198-310, 315-446, 449-618): the four functions are the same
    text with 4, 5, 6, 8 taps written out; coefficients live in int32 locals and are truncated to int16 only on exit
    (`coefN -= int32(int16(sgn))` adds a value in -1..1 to an int32)."""
    lim = order + 1
    if len(coefs) <= order - 1:  # _ = coefs[order-1]
        raise GoPanic("index out of range [%d] with length %d" % (order - 1, len(coefs)))
    if num > len(pc1) or num > len(out) or num < 0:
        raise GoPanic("slice bounds out of range [:%d]" % num)
    coef = [int(coefs[j]) for j in range(order)]  # coef0..coef{order-1} := int32(coefs[j])
    for idx in range(lim, num):
        w = out[idx - lim:idx]  # w[0] = top, w[lim-1] = out[idx-1]
        top = w[0]
        diff = [i32(top - w[lim - 1 - j]) for j in range(order)]  # diff0 = top - w[order], ..., diff{order-1} = top - w[1]
        acc = denHalf
        for j in range(order):  # denHalf - coef0*diff0 - coef1*diff1 - ...
            acc = i32(acc - i32(coef[j] * diff[j]))
        sum1 = sar32(acc, denShift)
        dl = pc1[idx]
        del0 = dl
        sign = signOfInt(dl)
        dl = i32(dl + top + sum1)
        out[idx] = sar32(shl_i32(dl, chanShift), chanShift)
        if sign > 0:
            done = False
            for j in range(order - 1, 0, -1):  # taps order-1 .. 1 with the early exit
                sgn = signOfInt(diff[j])
                coef[j] = i32(coef[j] - i16(sgn))
                del0 = i32(del0 - i32((order - j) * sar32(i32(sgn * diff[j]), denShift)))
                if del0 <= 0:
                    done = True
                    break
            if not done:
                coef[0] = i32(coef[0] - i16(signOfInt(diff[0])))
        elif sign < 0:
            done = False
            for j in range(order - 1, 0, -1):
                sgn = i32(-signOfInt(diff[j]))
                coef[j] = i32(coef[j] - i16(sgn))
                del0 = i32(del0 - i32((order - j) * sar32(i32(sgn * diff[j]), denShift)))
                if del0 >= 0:
                    done = True
                    break
            if not done:
                coef[0] = i32(coef[0] + i16(signOfInt(diff[0])))
        if trace is not None:
            trace.append(("fix", idx, top, list(diff), sum1, pc1[idx], out[idx], list(coef), del0))
    for j in range(order):
        coefs[j] = i16(coef[j])


def unpcBlockGeneral(pc1, out, num, coefs, numActive, lim, chanShift, denShift, denHalf, trace=None):  # predictor.go:623-684
    activeCount = numActive
    if activeCount > len(coefs):
        raise GoPanic("slice bounds out of range [:%d] with capacity %d" % (activeCount, len(coefs)))
    if num > len(pc1) or num > len(out) or num < 0:
        raise GoPanic("slice bounds out of range [:%d]" % num)
    coefsNA = coefs  # in-place int16 storage
    for idx in range(lim, num):
        hist = out[idx - lim:idx]
        top = hist[0]
        sum1 = 0
        for k in range(activeCount):
            sum1 = i32(sum1 + i32(coefsNA[k] * i32(hist[activeCount - k] - top)))
        dl = pc1[idx]
        del0 = dl
        sign = signOfInt(dl)
        dl = i32(dl + top + sar32(i32(sum1 + denHalf), denShift))
        out[idx] = sar32(shl_i32(dl, chanShift), chanShift)
        if sign > 0:
            for k in range(activeCount - 1, -1, -1):
                dd = i32(top - hist[activeCount - k])
                sgn = signOfInt(dd)
                coefsNA[k] = i16(coefsNA[k] - i16(sgn))
                del0 = i32(del0 - i32((activeCount - k) * sar32(i32(sgn * dd), denShift)))
                if del0 <= 0:
                    break
        elif sign < 0:
            for k in range(activeCount - 1, -1, -1):
                dd = i32(top - hist[activeCount - k])
                sgn = signOfInt(dd)
                coefsNA[k] = i16(coefsNA[k] + i16(sgn))
                del0 = i32(del0 - i32((activeCount - k) * sar32(i32(-sgn * dd), denShift)))
                if del0 >= 0:
                    break
        if trace is not None:
            trace.append(("gen", idx, top, sum1, pc1[idx], out[idx], list(coefsNA[:activeCount]), del0))


# ---- internal/alac/matrix.go + format.go ------------------------------------------------------------------------------
def BytesPerSample(depth):  # format.go:23-34
    if depth == 16:
        return 2
    if depth in (20, 24):
        return 3
    if depth == 32:
        return 4
    raise GoPanic("alac: BytesPerSample called with unsupported bit depth %d" % depth)


def _put(out, off, nbytes, val):
    if off < 0 or off + nbytes > len(out):
        raise GoPanic("slice bounds out of range [%d:%d] with capacity %d" % (off, off + nbytes, len(out)))
    v = u32(val)
    for b in range(nbytes):
        out[off + b] = (v >> (8 * b)) & 0xFF


def WriteStereo(depth, out, mixU, mixV, chanIdx, numChan, numSamples, mixBits, mixRes, shiftBuf, bytesShifted):
    """WriteStereo16/20/24/32, matrix.go:30-215."""
    bps = BytesPerSample(depth)
    stride = numChan * bps
    shift = bytesShifted * 8
    off = chanIdx * bps
    if numSamples > len(mixU) or numSamples > len(mixV):
        raise GoPanic("slice bounds out of range [:%d]" % numSamples)
    merge = depth in (24, 32) and bytesShifted != 0
    if merge and numSamples * 2 > len(shiftBuf):
        raise GoPanic("slice bounds out of range [:%d]" % (numSamples * 2))
    for idx in range(numSamples):
        if mixRes != 0:
            left = i32(mixU[idx] + mixV[idx] - sar32(i32(mixRes * mixV[idx]), mixBits))
            right = i32(left - mixV[idx])
        else:
            left = mixU[idx]
            right = mixV[idx]
        if depth == 20:
            left = shl_i32(left, 4)
            right = shl_i32(right, 4)
        if merge:
            left = i32(shl_i32(left, shift) | shiftBuf[idx * 2 + 0])
            right = i32(shl_i32(right, shift) | shiftBuf[idx * 2 + 1])
        if off < 0 or off + 2 * bps > len(out):
            raise GoPanic("slice bounds out of range [%d:%d] with capacity %d" % (off, off + 2 * bps, len(out)))
        _put(out, off, bps, left)
        _put(out, off + bps, bps, right)
        off += stride


def WriteMono(depth, out, mixU, chanIdx, numChan, numSamples, shiftBuf, bytesShifted):
    """WriteMono16/20/24/32, matrix.go:220-301."""
    bps = BytesPerSample(depth)
    stride = numChan * bps
    shift = bytesShifted * 8
    off = chanIdx * bps
    if numSamples > len(mixU):
        raise GoPanic("slice bounds out of range [:%d]" % numSamples)
    merge = depth in (24, 32) and bytesShifted != 0
    if merge and numSamples > len(shiftBuf):
        raise GoPanic("slice bounds out of range [:%d]" % numSamples)
    for idx in range(numSamples):
        val = mixU[idx]
        if depth == 20:
            val = shl_i32(val, 4)
        if merge:
            val = i32(shl_i32(val, shift) | shiftBuf[idx])
        _put(out, off, bps, val)
        off += stride


# ---- decoder.go ----------------------------------------------------------------------------------------------------
channelLayoutOffsets = [  # decoder.go:55-64
    [0], [0, 1], [2, 0, 1], [2, 0, 1, 3], [2, 0, 1, 3, 4], [2, 0, 1, 4, 5, 3], [2, 0, 1, 4, 5, 6, 3],
    [2, 6, 7, 0, 1, 4, 5, 3],
]
elemSCE, elemCPE, elemCCE, elemLFE, elemDSE, elemPCE, elemFIL, elemEND = range(8)


class PacketConfig:
    def __init__(self, FrameLength, BitDepth, NumChannels, PB=40, MB=10, KB=14, MaxRun=255):
        self.FrameLength, self.BitDepth, self.NumChannels = FrameLength, BitDepth, NumChannels
        self.PB, self.MB, self.KB, self.MaxRun = PB, MB, KB, MaxRun


class PacketDecoder:  # decoder.go:79-109
    def __init__(self, config, trace=None):
        if config.BitDepth not in (16, 20, 24, 32):
            raise ValueError("invalid configuration: alac: unsupported bit depth: %d" % config.BitDepth)
        self.config = config
        frameLen = config.FrameLength
        self.mixBufferU = [0] * frameLen
        self.mixBufferV = [0] * frameLen
        self.predictor = [0] * frameLen
        self.shiftBuffer = [0] * (frameLen * 2)
        self.bits = BitBuffer()
        self.trace = trace
        self.cpe_last_slot = False

    def DecodePacket(self, packet):  # decoder.go:117-128
        numChan = self.config.NumChannels
        bps = BytesPerSample(self.config.BitDepth)
        output = bytearray(self.config.FrameLength * numChan * bps)
        n = self.decodePacketInto(packet, output)
        return bytes(output[:n])

    def decodePacketInto(self, packet, output):  # decoder.go:133-207
        self.bits.Reset(packet)
        bits = self.bits
        numSamples = self.config.FrameLength
        numChan = self.config.NumChannels
        bps = BytesPerSample(self.config.BitDepth)
        chanIdx = 0
        if numChan < 1 or numChan > 8:
            raise GoPanic("index out of range [%d] with length 8" % (numChan - 1))
        offsets = channelLayoutOffsets[numChan - 1]
        while True:
            if bits.PastEnd():
                raise DecodeError(ErrBitstreamOverrun)
            tag = bits.ReadSmall(3)
            if tag in (elemSCE, elemLFE):
                if chanIdx >= len(offsets):
                    raise GoPanic("index out of range")
                outChanIdx = offsets[chanIdx]
                try:
                    ns = self.decodeSCE(bits, output, outChanIdx, numChan, numSamples)
                except DecodeError as e:
                    raise DecodeError("SCE/LFE", *e.chain)
                numSamples = ns
                chanIdx += 1
            elif tag == elemCPE:
                if chanIdx + 2 > numChan:
                    break
                outChanIdx = offsets[chanIdx]
                if outChanIdx + 2 > numChan:
                    # the pair does not fit the frame: WriteStereo* writes past it (and panics only on a full frame);
                    # the product calls this malformed up front (documented deviation)
                    self.cpe_last_slot = True
                try:
                    ns = self.decodeCPE(bits, output, outChanIdx, numChan, numSamples)
                except DecodeError as e:
                    raise DecodeError("CPE", *e.chain)
                numSamples = ns
                chanIdx += 2
            elif tag in (elemCCE, elemPCE):
                raise DecodeError(ErrUnsupportedElement)
            elif tag == elemDSE:
                try:
                    self.skipDSE(bits)
                except DecodeError as e:
                    raise DecodeError("DSE", *e.chain)
            elif tag == elemFIL:
                try:
                    self.skipFIL(bits)
                except DecodeError as e:
                    raise DecodeError("FIL", *e.chain)
            elif tag == elemEND:
                bits.ByteAlign()
                break
            if chanIdx >= numChan:
                break
        return numSamples * numChan * bps

    def _header(self, bits, cpe):  # decoder.go:213-235 / 351-376
        bits.ReadSmall(4)
        unusedHeader = bits.Read(12)
        if unusedHeader != 0:
            raise DecodeError(ErrInvalidHeader)
        headerByte = bits.Read(4)
        partialFrame = headerByte >> 3
        bytesShifted = (headerByte >> 1) & 0x3
        if bytesShifted == 3:
            raise DecodeError(ErrInvalidShift)
        escapeFlag = headerByte & 0x1
        chanBits = u32(self.config.BitDepth - bytesShifted * 8 + (1 if cpe else 0))
        return partialFrame, bytesShifted, escapeFlag, chanBits

    def decodeSCE(self, bits, output, chanIdx, numChan, numSamples):  # decoder.go:210-265
        partialFrame, bytesShifted, escapeFlag, chanBits = self._header(bits, False)
        if partialFrame != 0:
            numSamples = shl32(bits.Read(16), 16)
            numSamples |= bits.Read(16)
        if escapeFlag == 0:
            self.decodeSCECompressed(bits, chanBits, bytesShifted, numSamples)
        else:
            self.decodeSCEEscape(bits, chanBits, numSamples)
            bytesShifted = 0
        WriteMono(self.config.BitDepth, output, self.mixBufferU, chanIdx, numChan, numSamples, self.shiftBuffer,
                  bytesShifted)
        return numSamples

    def _chan_header(self, bits):  # decoder.go:275-286
        headerByte = bits.Read(8)
        mode = headerByte >> 4
        denShift = headerByte & 0xF
        headerByte = bits.Read(8)
        pbFactor = headerByte >> 5
        num = headerByte & 0x1F
        coefs = [0] * MAX_COEFS
        for i in range(num):
            coefs[i] = i16(bits.Read(16))
        return mode, denShift, pbFactor, num, coefs

    def _entropy_predict(self, bits, chanBits, numSamples, hdr, mixBuf, stage):
        mode, denShift, pbFactor, num, coefs = hdr
        predBound = self.config.PB
        agP = AGParams(self.config.MB, (predBound * pbFactor) // 4, self.config.KB, numSamples, numSamples,
                       self.config.MaxRun)
        try:
            DynDecomp(agP, bits, self.predictor, numSamples, chanBits, self.trace)
        except DecodeError as e:
            raise DecodeError(stage, *e.chain)
        if mode != 0:  # decoder.go:307-309
            UnpcBlock(self.predictor, self.predictor, numSamples, None, NUM_ACTIVE_DELTA, chanBits, 0, self.trace)
        UnpcBlock(self.predictor, mixBuf, numSamples, coefs[:num], num, chanBits, denShift, self.trace)

    def decodeSCECompressed(self, bits, chanBits, bytesShifted, numSamples):  # decoder.go:267-324
        bits.Read(8)
        bits.Read(8)
        hdr = self._chan_header(bits)
        shiftBits = None
        if bytesShifted != 0:
            shiftBits = bits.Copy()
            bits.Advance(u32(u32(bytesShifted * 8) * u32(numSamples)))
        self._entropy_predict(bits, chanBits, numSamples, hdr, self.mixBufferU, "entropy decode")
        if bytesShifted != 0:
            shift = (bytesShifted * 8) & 0xFF
            if numSamples > len(self.shiftBuffer):
                raise GoPanic("slice bounds out of range [:%d]" % numSamples)
            for i in range(numSamples):
                self.shiftBuffer[i] = shiftBits.Read(shift) & 0xFFFF

    def decodeSCEEscape(self, bits, chanBits, numSamples):  # decoder.go:326-345
        shift = u32(32 - chanBits)
        if numSamples > len(self.mixBufferU):
            raise GoPanic("slice bounds out of range [:%d]" % numSamples)
        if chanBits <= 16:
            for idx in range(numSamples):
                val = i32(bits.Read(chanBits & 0xFF))
                self.mixBufferU[idx] = sar32(shl_i32(val, shift), shift)
        else:
            extraBits = chanBits - 16
            for idx in range(numSamples):
                val = i32(bits.Read(16))
                val = sar32(shl_i32(val, 16), shift)
                self.mixBufferU[idx] = i32(val | i32(bits.Read(extraBits & 0xFF)))

    def decodeCPE(self, bits, output, chanIdx, numChan, numSamples):  # decoder.go:348-414
        partialFrame, bytesShifted, escapeFlag, chanBits = self._header(bits, True)
        if partialFrame != 0:
            numSamples = shl32(bits.Read(16), 16)
            numSamples |= bits.Read(16)
        mixBits = mixRes = 0
        if escapeFlag == 0:
            mixBits, mixRes = self.decodeCPECompressed(bits, chanBits, bytesShifted, numSamples)
        else:
            chanBits = self.config.BitDepth
            self.decodeCPEEscape(bits, chanBits, numSamples)
            bytesShifted = 0
        WriteStereo(self.config.BitDepth, output, self.mixBufferU, self.mixBufferV, chanIdx, numChan, numSamples,
                    mixBits, mixRes, self.shiftBuffer, bytesShifted)
        return numSamples

    def decodeCPECompressed(self, bits, chanBits, bytesShifted, numSamples):  # decoder.go:416-505
        mixBits = i32(bits.Read(8))
        mixRes = i8(bits.Read(8))
        hdrU = self._chan_header(bits)
        hdrV = self._chan_header(bits)
        shiftBits = None
        if bytesShifted != 0:
            shiftBits = bits.Copy()
            bits.Advance(u32(u32(bytesShifted * 8 * 2) * u32(numSamples)))
        self._entropy_predict(bits, chanBits, numSamples, hdrU, self.mixBufferU, "entropy decode U")
        self._entropy_predict(bits, chanBits, numSamples, hdrV, self.mixBufferV, "entropy decode V")
        if bytesShifted != 0:
            shift = (bytesShifted * 8) & 0xFF
            if numSamples * 2 > len(self.shiftBuffer):
                raise GoPanic("slice bounds out of range [:%d]" % (numSamples * 2))
            for i in range(numSamples):
                self.shiftBuffer[2 * i] = shiftBits.Read(shift) & 0xFFFF
                self.shiftBuffer[2 * i + 1] = shiftBits.Read(shift) & 0xFFFF
        return mixBits, mixRes

    def decodeCPEEscape(self, bits, chanBits, numSamples):  # decoder.go:507-535
        shift = u32(32 - chanBits)
        if numSamples > len(self.mixBufferU):
            raise GoPanic("slice bounds out of range [:%d]" % numSamples)
        for idx in range(numSamples):
            for buf in (self.mixBufferU, self.mixBufferV):
                if chanBits <= 16:
                    val = i32(bits.Read(chanBits & 0xFF))
                    buf[idx] = sar32(shl_i32(val, shift), shift)
                else:
                    val = i32(bits.Read(16))
                    val = sar32(shl_i32(val, 16), shift)
                    buf[idx] = i32(val | i32(bits.Read((chanBits - 16) & 0xFF)))

    def skipFIL(self, bits):  # decoder.go:538-552
        count = i16(bits.ReadSmall(4))
        if count == 15:
            count = i16(count + i16(bits.ReadSmall(8)) - 1)
        bits.Advance(u32(u32(count) * 8))
        if bits.PastEnd():
            raise DecodeError(ErrBitstreamOverrun)

    def skipDSE(self, bits):  # decoder.go:555-574
        bits.ReadSmall(4)
        dataByteAlignFlag = bits.ReadOne()
        count = bits.ReadSmall(8)
        if count == 255:
            count = (count + bits.ReadSmall(8)) & 0xFFFF
        if dataByteAlignFlag != 0:
            bits.ByteAlign()
        bits.Advance(u32(count * 8))
        if bits.PastEnd():
            raise DecodeError(ErrBitstreamOverrun)


# ---- status word of include/alacgpu.h for comparisons with the oracle / the GPU ------------------------------------
_CODE = {ErrBitstreamOverrun: 1, ErrSampleOverrun: 2, ErrInvalidHeader: 3, ErrInvalidShift: 4, ErrUnsupportedElement: 5}
_CTX = {"SCE/LFE": 1, "CPE": 2, "DSE": 3, "FIL": 4}
_STAGE = {"entropy decode": 1, "entropy decode U": 2, "entropy decode V": 3}


def decode_packet(config, packet, trace=None, info=None):
    """-> (pcm bytes, frames, status word) with the conventions of include/alacgpu.h (a Go panic = code 6).
    info (dict, optional) receives 'cpe_last_slot' (see decodePacketInto)."""
    dec = PacketDecoder(config, trace)
    try:
        return _decode_packet(dec, config, packet)
    finally:
        if info is not None:
            info["cpe_last_slot"] = dec.cpe_last_slot


def _decode_packet(dec, config, packet):
    try:
        pcm = dec.DecodePacket(packet)
    except GoPanic:
        return b"", 0, 6
    except DecodeError as e:
        code = _CODE[e.chain[-1]]
        ctx = stage = 0
        for c in e.chain[:-1]:
            ctx = _CTX.get(c, ctx)
            stage = _STAGE.get(c, stage)
        return b"", 0, code | (ctx << 8) | (stage << 12)
    bps = BytesPerSample(config.BitDepth)
    return pcm, len(pcm) // (config.NumChannels * bps), 0
