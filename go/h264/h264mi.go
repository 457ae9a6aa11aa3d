// Package h264 -- cgo shim that keeps the exported names of mrmod/h264decode's package h264
// (h264/nalUnit.go, h264/sps.go, h264/pps.go, h264/slice.go) on top of libh264mi.so.
//
// UNVERIFIED: the build image has no Go toolchain; this file was written against the cgo rules and
// the C header include/h264mi.h but has never been compiled.  The same ABI is exercised from
// Python ctypes by the test-suite.  The parameter-set and slice-header structs carry EVERY field of the
// C structs (structs_gen.go, generated from the header by tools/gen_go_structs.py).
package h264

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../h264decode_amd -lh264mi -Wl,-rpath,${SRCDIR}/../../h264decode_amd
#include <stdlib.h>
#include "h264mi.h"
*/
import "C"

import (
	"fmt"
	"log"
	"os"
	"unsafe"
)

var logger = log.New(os.Stderr, "streamer ", log.Lshortfile|log.Lmicroseconds) // h264/server.go:25-27

func status(rc C.int32_t) error {
	if rc == 0 {
		return nil
	}
	return fmt.Errorf("h264mi %d: %s", int(rc), C.GoString(C.h264mi_last_error_string()))
}

func bptr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// NalUnit mirrors h264/nalUnit.go:3-30 (fields the C ABI carries).
type NalUnit struct {
	NumBytes, ForbiddenZeroBit, RefIdc, Type     int
	SvcExtensionFlag, Avc3dExtensionFlag         int
	HeaderBytes                                  int
	rbsp                                         []byte
}

func (n *NalUnit) RBSP() []byte { return n.rbsp } // h264/nalUnit.go:72

// NewNalUnit: h264/nalUnit.go:75.
func NewNalUnit(frame []byte, numBytesInNal int) *NalUnit {
	var c C.h264mi_nal
	rbsp := make([]byte, numBytesInNal)
	var rl C.size_t
	if err := status(C.h264mi_nal_parse(bptr(frame), C.size_t(numBytesInNal), &c, bptr(rbsp), &rl)); err != nil {
		logger.Printf("NewNalUnit: %v", err)
		return &NalUnit{}
	}
	return &NalUnit{NumBytes: int(c.num_bytes), ForbiddenZeroBit: int(c.forbidden_zero_bit), RefIdc: int(c.ref_idc), Type: int(c._type),
		SvcExtensionFlag: int(c.svc_extension_flag), Avc3dExtensionFlag: int(c.avc_3d_extension_flag), HeaderBytes: int(c.header_bytes), rbsp: rbsp[:int(rl)]}
}

// ReadNalUnits replaces the readNalUnit loop (h264/server.go:64-111).
func ReadNalUnits(stream []byte) ([]*NalUnit, error) {
	capN := 1024
	for {
		arr := make([]C.h264mi_nal, capN)
		var n C.int32_t
		rc := C.h264mi_annexb_scan(bptr(stream), C.size_t(len(stream)), &arr[0], C.int32_t(capN), &n)
		if rc == C.H264MI_ECAPACITY {
			capN *= 4
			continue
		}
		if err := status(rc); err != nil {
			return nil, err
		}
		out := make([]*NalUnit, 0, int(n))
		for i := 0; i < int(n); i++ {
			off, size := int(arr[i].offset), int(arr[i].num_bytes)
			out = append(out, NewNalUnit(stream[off:off+size], size))
		}
		return out, nil
	}
}

// SPS mirrors h264/sps.go:9-103: every field the C ABI carries, under the reference's names (SPSFields, structs_gen.go).
type SPS struct {
	c C.h264mi_sps
	SPSFields
}

func NewSPS(rbsp []byte, showPacket bool) *SPS { // h264/sps.go:192
	s := &SPS{}
	if err := status(C.h264mi_sps_parse(bptr(rbsp), C.size_t(len(rbsp)), &s.c)); err != nil {
		logger.Printf("NewSPS: %v", err)
		return s
	}
	s.SPSFields = copySPSFields(&s.c)
	return s
}

// PPS mirrors h264/pps.go:10-38 (PPSFields, structs_gen.go).
type PPS struct {
	c C.h264mi_pps
	PPSFields
	SliceGroupId []int // h264/pps.go:23: slice_group_map_type 6, one entry per map unit
	ids          []byte
}

func NewPPS(sps *SPS, rbsp []byte, showPacket bool) *PPS { // h264/pps.go:40
	p := &PPS{}
	if err := status(C.h264mi_pps_parse(&sps.c, bptr(rbsp), C.size_t(len(rbsp)), &p.c)); err != nil {
		logger.Printf("NewPPS: %v", err)
		return p
	}
	p.PPSFields = copyPPSFields(&p.c)
	if p.NumSliceGroupsMinus1 > 0 && p.SliceGroupMapType == 6 {
		p.ids = make([]byte, p.PicSizeInMapUnitsMinus1+1)
		var n C.size_t
		if err := status(C.h264mi_pps_slice_group_ids(&sps.c, bptr(rbsp), C.size_t(len(rbsp)), bptr(p.ids), C.size_t(len(p.ids)), &n)); err != nil {
			logger.Printf("NewPPS: %v", err)
			return p
		}
		for _, v := range p.ids {
			p.SliceGroupId = append(p.SliceGroupId, int(v))
		}
	}
	return p
}

// ---- slice groups, 8.2.2 (h264/slice.go:134-158 MbToSliceGroupMap, :457-529 MapUnitToSliceGroupMap, :530-552 nextMbAddress) ----
func sliceGroupCycle(header *SliceHeader) C.int32_t {
	if header == nil {
		return 0
	}
	return C.int32_t(header.SliceGroupChangeCycle)
}

// MapUnitToSliceGroupMap: all seven map types (the reference stops at type 2).
func MapUnitToSliceGroupMap(sps *SPS, pps *PPS, header *SliceHeader) []int {
	buf := make([]byte, (sps.PicWidthInMbsMinus1+1)*(sps.PicHeightInMapUnitsMinus1+1))
	if err := status(C.h264mi_map_unit_to_slice_group_map(&sps.c, &pps.c, bptr(pps.ids), C.size_t(len(pps.ids)), sliceGroupCycle(header), bptr(buf), C.size_t(len(buf)), nil)); err != nil {
		logger.Printf("MapUnitToSliceGroupMap: %v", err)
		return nil
	}
	out := make([]int, len(buf))
	for i, v := range buf {
		out[i] = int(v)
	}
	return out
}

func MbToSliceGroupMap(sps *SPS, pps *PPS, header *SliceHeader) []int {
	field := 0
	if header != nil && header.FieldPic != 0 {
		field = 1
	}
	n := (sps.PicWidthInMbsMinus1 + 1) * (sps.PicHeightInMapUnitsMinus1 + 1)
	if sps.FrameMbsOnly == 0 && field == 0 {
		n *= 2
	}
	buf := make([]byte, n)
	if err := status(C.h264mi_mb_to_slice_group_map(&sps.c, &pps.c, bptr(pps.ids), C.size_t(len(pps.ids)), sliceGroupCycle(header), C.int32_t(field), bptr(buf), C.size_t(len(buf)), nil)); err != nil {
		logger.Printf("MbToSliceGroupMap: %v", err)
		return nil
	}
	out := make([]int, len(buf))
	for i, v := range buf {
		out[i] = int(v)
	}
	return out
}

// nextMbAddress: the reference's signature (h264/slice.go:530); PicSizeInMbs when n is the last macroblock of its group.
func nextMbAddress(n int, sps *SPS, pps *PPS, header *SliceHeader) int {
	m := MbToSliceGroupMap(sps, pps, header)
	i := n + 1
	for i < len(m) && m[i] != m[n] {
		i++
	}
	return i
}

// SliceHeader mirrors h264/slice.go:23-75 incl. the list-1 / direct / weighted-prediction fields of B slices (SliceHeaderFields).
type SliceHeader struct{ SliceHeaderFields }
type Slice struct{ Header *SliceHeader }
type VideoStream struct { // h264/slice.go:8-12
	SPS    *SPS
	PPS    *PPS
	Slices []*SliceContext
}
type SliceContext struct { // h264/slice.go:13-18
	*NalUnit
	*SPS
	*PPS
	*Slice
}

func NewSliceContext(vs *VideoStream, nal *NalUnit, rbsp []byte, showPacket bool) *SliceContext { // h264/slice.go:835
	var c C.h264mi_slice_header
	if err := status(C.h264mi_slice_header_parse(&vs.SPS.c, &vs.PPS.c, C.int32_t(nal.RefIdc), C.int32_t(nal.Type), bptr(rbsp), C.size_t(len(rbsp)), &c)); err != nil {
		logger.Printf("NewSliceContext: %v", err)
		return &SliceContext{NalUnit: nal, SPS: vs.SPS, PPS: vs.PPS, Slice: &Slice{Header: &SliceHeader{}}}
	}
	return &SliceContext{NalUnit: nal, SPS: vs.SPS, PPS: vs.PPS, Slice: &Slice{Header: &SliceHeader{copySliceHeaderFields(&c)}}}
}

// ---- macroblock layer (h264/slice.go:77-102 SliceData, :570 NewSliceData, h264/mbType.go:75 MbTypeName) ----
// slice_data() is decoded by the GPU entropy kernel; NewSliceData reads the records it left (128 bytes per macroblock,
// h264decode_amd/csrc/mi_types.h MbRec) back through h264mi_frame_read_mbrecs / h264mi_frame_read_mbmv1.

const MbTypeInferred = 1000 // h264/mbType.go:5 MB_TYPE_INFERRED

type SliceData struct {
	MbType                                    int // as coded: Tables 7-11 / 7-13 / 7-14 (intra types in P / B slices: + 5 / + 23)
	MbTypeName                                string
	MbSkipFlag, TransformSize8x8Flag          bool
	QPY, CodedBlockPattern, IntraChromaPredMode int
	Intra4x4PredMode                          []int8
	SubMbType                                 []int8
	RefIdxL0, RefIdxL1                        []int8
	MvL0, MvL1                                [][2]int16 // final vectors per 4x4 block (the kernel adds the prediction of 8.4.1.3)
}

// MbTypeName: h264/mbType.go:75-88.  The names follow the rule of Tables 7-11 / 7-13 / 7-14; only the frequent ones are spelled
// out here, the complete tables live in the Python mirror (h264decode_amd/mbtype.py), which the tests check against the reference.
func MbTypeName(sliceType string, mbType int) string {
	if mbType == MbTypeInferred {
		if sliceType == "B" {
			return "B_Skip"
		}
		return "P_Skip"
	}
	off := map[string]int{"I": 0, "P": 5, "SP": 5, "B": 23}[sliceType]
	if mbType >= off {
		switch it := mbType - off; {
		case it == 0:
			return "I_NxN"
		case it == 25:
			return "I_PCM"
		default:
			return fmt.Sprintf("I_16x16_%d_%d_%d", (it-1)&3, ((it-1)>>2)%3, (it-1)/12)
		}
	}
	if sliceType == "B" {
		return [...]string{"B_Direct_16x16", "B_L0_16x16", "B_L1_16x16", "B_Bi_16x16", "B_L0_L0_16x8", "B_L0_L0_8x16", "B_L1_L1_16x8", "B_L1_L1_8x16",
			"B_L0_L1_16x8", "B_L0_L1_8x16", "B_L1_L0_16x8", "B_L1_L0_8x16", "B_L0_Bi_16x8", "B_L0_Bi_8x16", "B_L1_Bi_16x8", "B_L1_Bi_8x16",
			"B_Bi_L0_16x8", "B_Bi_L0_8x16", "B_Bi_L1_16x8", "B_Bi_L1_8x16", "B_Bi_Bi_16x8", "B_Bi_Bi_8x16", "B_8x8"}[mbType]
	}
	return [...]string{"P_L0_16x16", "P_L0_L0_16x8", "P_L0_L0_8x16", "P_8x8", "P_8x8ref0"}[mbType]
}

// NewSliceData: h264/slice.go:570, with the Decoder that decoded the picture in the place of the bit reader.
func NewSliceData(sc *SliceContext, d *Decoder, stream, frame int) ([]SliceData, error) {
	n := int(sc.SPS.PicWidthInMbs) * int(sc.SPS.PicHeightInMbs)
	rec := make([]byte, n*128)
	if err := status(C.h264mi_frame_read_mbrecs(d.h, C.int32_t(stream), C.int32_t(frame), bptr(rec), C.size_t(len(rec)))); err != nil {
		return nil, err
	}
	st := [...]string{"P", "B", "I", "SP", "SI"}[sc.Slice.Header.SliceType%5]
	var mv1 []byte
	if st == "B" {
		mv1 = make([]byte, n*64)
		if err := status(C.h264mi_frame_read_mbmv1(d.h, C.int32_t(stream), C.int32_t(frame), bptr(mv1), C.size_t(len(mv1)))); err != nil {
			return nil, err
		}
	}
	i16 := func(b []byte, i int) int16 { return int16(uint16(b[2*i]) | uint16(b[2*i+1])<<8) }
	out := make([]SliceData, n)
	for m := range out {
		r := rec[m*128 : m*128+128]
		t, cbp := int(r[0]), int(r[5])
		sd := SliceData{TransformSize8x8Flag: r[1] != 0, QPY: int(r[2]), CodedBlockPattern: cbp, IntraChromaPredMode: int(r[6]), MbSkipFlag: t == 9 || t == 12}
		intra, inter := t >= 1 && t <= 4, t >= 5
		raw := 0
		switch {
		case t == 3:
			raw = 1 + int(r[7]) + 4*(cbp>>4)
			if cbp&15 != 0 {
				raw += 12
			}
		case t == 4:
			raw = 25
		case sd.MbSkipFlag:
			raw = MbTypeInferred
		case inter:
			raw = int(r[20]) // MbRec.ipm[4]: mb_type as coded
		}
		if intra {
			raw += map[string]int{"P": 5, "SP": 5, "B": 23}[st]
		}
		sd.MbType, sd.MbTypeName = raw, MbTypeName(st, raw)
		if t == 1 || t == 2 {
			for k := 0; k < 16; k++ {
				sd.Intra4x4PredMode = append(sd.Intra4x4PredMode, int8(r[16+k]))
			}
		}
		if inter {
			for k := 0; k < 4; k++ {
				sd.RefIdxL0 = append(sd.RefIdxL0, int8(r[32+k]))
			}
			for k := 0; k < 16; k++ {
				sd.MvL0 = append(sd.MvL0, [2]int16{i16(r[48:], 2*k), i16(r[48:], 2*k+1)})
			}
			if !sd.MbSkipFlag && (raw == 3 && st != "B" || raw == 22 && st == "B") {
				for k := 0; k < 4; k++ {
					sd.SubMbType = append(sd.SubMbType, int8(r[21+k]))
				}
			}
			if st == "B" {
				for k := 0; k < 4; k++ {
					ref := int8(-1)
					if i16(r[120:], k) >= 0 {
						ref = int8(r[16+k])
					}
					sd.RefIdxL1 = append(sd.RefIdxL1, ref)
				}
				for k := 0; k < 16; k++ {
					sd.MvL1 = append(sd.MvL1, [2]int16{i16(mv1[m*64:], 2*k), i16(mv1[m*64:], 2*k+1)})
				}
			}
		}
		out[m] = sd
	}
	return out, nil
}

// ---- additive API: batched GPU decode (the reference has no pixel type) ----

type Config struct {
	Device, MaxStreams, MaxWidth, MaxHeight, MaxFramesPerBatch, MaxSlicesPerFrame int
	MaxBitstreamBytes                                                              int64
	MaxRefFrames, CoefBlocksPerMb                                                  int // 0 = defaults (16 reference slots per stream, 8 residual blocks per macroblock)
	BPictures                                                                      int // 1 = the buffers only B pictures need exist from the start (h264mi_config.b_pictures)
	AllowUnpinnedFieldCabac                                                        int // 1 = CABAC field pictures are decoded with the unpinned context tables (h264mi_config.allow_unpinned_field_cabac)
}
type Decoder struct{ h *C.h264mi_decoder }
type BatchInfo struct {
	Frames, Slices             int
	Width, Height              int
	CodedWidth, CodedHeight    int
}

func NewDecoder(cfg Config) (*Decoder, error) {
	c := C.h264mi_config{struct_size: C.uint32_t(C.sizeof_h264mi_config), device: C.int32_t(cfg.Device), max_streams: C.int32_t(cfg.MaxStreams), max_width: C.int32_t(cfg.MaxWidth),
		max_height: C.int32_t(cfg.MaxHeight), max_frames_per_batch: C.int32_t(cfg.MaxFramesPerBatch),
		max_slices_per_frame: C.int32_t(cfg.MaxSlicesPerFrame), max_bitstream_bytes: C.int64_t(cfg.MaxBitstreamBytes),
		max_ref_frames: C.int32_t(cfg.MaxRefFrames), coef_blocks_per_mb: C.int32_t(cfg.CoefBlocksPerMb), b_pictures: C.int32_t(cfg.BPictures), allow_unpinned_field_cabac: C.int32_t(cfg.AllowUnpinnedFieldCabac)}
	d := &Decoder{}
	if err := status(C.h264mi_decoder_create(&c, &d.h)); err != nil {
		return nil, err
	}
	return d, nil
}
func (d *Decoder) Close() { C.h264mi_decoder_destroy(d.h); d.h = nil }

// DecodeBatch: one Annex-B chunk (whole access units) per stream.  The chunks are copied into C
// memory for the duration of the call (cgo forbids passing Go memory that holds Go pointers).
func (d *Decoder) DecodeBatch(chunks [][]byte) (BatchInfo, error) {
	n := len(chunks)
	ptrs := (*[1 << 28]*C.uint8_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	lens := (*[1 << 28]C.size_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.size_t(0)))))[:n:n]
	defer C.free(unsafe.Pointer(&ptrs[0]))
	defer C.free(unsafe.Pointer(&lens[0]))
	for i, b := range chunks {
		ptrs[i] = (*C.uint8_t)(C.CBytes(b))
		lens[i] = C.size_t(len(b))
		defer C.free(unsafe.Pointer(ptrs[i]))
	}
	var info C.h264mi_batch_info
	if err := status(C.h264mi_decode_batch(d.h, C.int32_t(n), &ptrs[0], &lens[0], &info)); err != nil {
		return BatchInfo{}, err
	}
	return BatchInfo{Frames: int(info.n_frames), Slices: int(info.n_slices), Width: int(info.width), Height: int(info.height),
		CodedWidth: int(info.coded_width), CodedHeight: int(info.coded_height)}, nil
}

// FrameRead returns tight I420 (Y, Cb, Cr back to back).
func (d *Decoder) FrameRead(stream, frame int, crop bool, w, h int) ([]byte, error) {
	buf := make([]byte, w*h*3/2)
	cr := 0
	if crop {
		cr = 1
	}
	if err := status(C.h264mi_frame_read(d.h, C.int32_t(stream), C.int32_t(frame), C.int32_t(cr), bptr(buf), C.size_t(len(buf)))); err != nil {
		return nil, err
	}
	return buf, nil
}

// ---- round-2 API: per-stream control, per-frame geometry / picture order, display order, pipelined ingest ----
//
// Threading (include/h264mi.h "Threading"): a decoder handle is used by one goroutine at a time; the library selects the
// decoder's device per call, but hipSetDevice and the last-error string are per OS thread, so a goroutine that reads
// LastError after a failing call must not have migrated: wrap call + error fetch in runtime.LockOSThread /
// UnlockOSThread (status() above does both inside one cgo call sequence and is safe as long as the goroutine is locked).

type FrameInfo struct {
	Width, Height, CodedWidth, CodedHeight, CropX, CropY int
	PicOrderCnt, FrameNum, NalRefIdc                     int
	IDR, NewSequence                                     bool // NewSequence: picture order counts start over (IDR or MMCO 5)
}

// SetIsolation: a stream with a bitstream error leaves the batch (its status is reported by StreamStatus, it decodes again
// from its next IDR picture); the other streams of the batch are not affected.
func (d *Decoder) SetIsolation(on bool) error {
	v := 0
	if on {
		v = 1
	}
	return status(C.h264mi_decoder_set_isolation(d.h, C.int32_t(v)))
}
func (d *Decoder) StreamStatus(stream int) (int, error) {
	var st C.int32_t
	err := status(C.h264mi_stream_status(d.h, C.int32_t(stream), &st))
	return int(st), err
}

// StreamReset forgets parameter sets, reference pictures and POC history of one stream slot (a new connection takes it over).
func (d *Decoder) StreamReset(stream int) error { return status(C.h264mi_stream_reset(d.h, C.int32_t(stream))) }

func (d *Decoder) FrameCount(stream int) (int, error) {
	var n C.int32_t
	err := status(C.h264mi_stream_frame_count(d.h, C.int32_t(stream), &n))
	return int(n), err
}
func (d *Decoder) FrameInfo(stream, frame int) (FrameInfo, error) {
	var c C.h264mi_frame_info
	if err := status(C.h264mi_frame_get_info(d.h, C.int32_t(stream), C.int32_t(frame), &c)); err != nil {
		return FrameInfo{}, err
	}
	return FrameInfo{Width: int(c.width), Height: int(c.height), CodedWidth: int(c.coded_width), CodedHeight: int(c.coded_height), CropX: int(c.crop_x),
		CropY: int(c.crop_y), PicOrderCnt: int(c.pic_order_cnt), FrameNum: int(c.frame_num), NalRefIdc: int(c.nal_ref_idc), IDR: c.idr != 0, NewSequence: c.new_sequence != 0}, nil
}

// OutputOrder: indices (decoding order) of the stream's frames of the last batch in display order -- ascending PicOrderCnt
// per coded video sequence.  Differs from 0..n-1 only for streams with B pictures.
func (d *Decoder) OutputOrder(stream int) ([]int, error) {
	n, err := d.FrameCount(stream)
	if err != nil || n == 0 {
		return nil, err
	}
	buf := make([]C.int32_t, n)
	var got C.int32_t
	if err := status(C.h264mi_stream_output_order(d.h, C.int32_t(stream), &buf[0], C.int32_t(n), &got)); err != nil {
		return nil, err
	}
	out := make([]int, int(got))
	for i := range out {
		out[i] = int(buf[i])
	}
	return out, nil
}

// Prepare / Execute / Sync: the three phases of DecodeBatch.  Prepare(n+1) may be called while batch n is still executing
// (two staging sets): host parsing and the H2D copies then overlap the kernels of batch n.  The frames of batch n stay
// readable until the batch after n+1 is prepared.
func (d *Decoder) Execute() error { return status(C.h264mi_batch_execute(d.h)) }
func (d *Decoder) Sync() error    { return status(C.h264mi_batch_sync(d.h)) }

// PackBatch writes tight cropped I420 copies of every frame of the last batch (stream < 0: all streams, stream-major) into
// device memory `dst` with one kernel launch and returns the byte count.
func (d *Decoder) PackBatch(stream int, dst unsafe.Pointer, capBytes int) (int, error) {
	var n C.size_t
	err := status(C.h264mi_batch_pack_device(d.h, C.int32_t(stream), dst, C.size_t(capBytes), &n))
	return int(n), err
}
