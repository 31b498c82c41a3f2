/*
 * oracle/ref_encode.cc — TEST INFRASTRUCTURE ONLY.  A small driver around the reference's experimental encoder
 * (en265.h API) used to produce additional real HEVC bitstreams for recorded golden fixtures
 * (tests/golden/make_enc_fixture.py).  Never part of the product library.
 *
 * Why not the reference's own enc265 CLI: at this commit its input path (ImageSource_YUV::read_next_image,
 * image-io.cc:67-73, and en265_allocate_image, en265.cc:178-192) allocates the input picture with a null SPS and the
 * default allocator then dereferences it (image.cc:164 -> fill_plane -> get_bit_depth), so it SEGVs on the first frame.
 * This driver allocates the input picture with a stub 8-bit 4:2:0 SPS instead (the way the encoder allocates its own
 * pictures, encoder/encoder-core.cc:141) and otherwise follows enc265.cc:300-345: push image, en265_encode, drain packets.
 *
 * usage: ref_encode in.yuv W H nframes out.bin [encoder options as enc265 takes them, e.g. --sop-structure intra -q 30]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <vector>

#include "libde265/en265.h"
#include "libde265/image.h"
#include "libde265/sps.h"

int main(int argc, char** argv)
{
  if (argc < 6) {
    fprintf(stderr, "usage: %s in.yuv W H nframes out.bin [encoder options]\n", argv[0]);
    return 2;
  }
  const char* in_name = argv[1];
  const int W = atoi(argv[2]), H = atoi(argv[3]), N = atoi(argv[4]);
  const char* out_name = argv[5];

  de265_init();
  en265_encoder_context* ectx = en265_new_encoder();

  /* the remaining arguments go to the encoder's own option parser (argv[0] is skipped by it) */
  std::vector<char*> args;
  args.push_back(argv[0]);
  for (int i = 6; i < argc; i++) args.push_back(argv[i]);
  int nargs = (int)args.size();
  if (en265_parse_command_line_parameters(ectx, &nargs, args.data()) != DE265_OK) {
    fprintf(stderr, "bad encoder option\n");
    return 2;
  }

  FILE* fin = fopen(in_name, "rb");
  FILE* fout = fopen(out_name, "wb");
  if (!fin || !fout) { fprintf(stderr, "cannot open files\n"); return 2; }

  std::shared_ptr<seq_parameter_set> stub = std::make_shared<seq_parameter_set>();
  stub->set_defaults();
  stub->BitDepth_Y = stub->BitDepth_C = 8;
  stub->chroma_format_idc = 1;
  stub->ChromaArrayType = 1;
  stub->SubWidthC = stub->SubHeightC = 2;

  en265_start_encoder(ectx, 0);

  std::vector<uint8_t> row(W);
  bool eof = false;
  for (int poc = 0; poc <= N && !eof; poc++) {
    de265_image* img = nullptr;
    if (poc < N) {
      img = new de265_image;
      if (img->alloc_image(W, H, de265_chroma_420, stub, false, nullptr, poc, nullptr, false) != DE265_OK) return 3;
      bool ok = true;
      for (int c = 0; c < 3 && ok; c++) {
        const int w = c ? W / 2 : W, h = c ? H / 2 : H;
        uint8_t* p = img->get_image_plane(c);
        const int stride = img->get_image_stride(c);
        for (int y = 0; y < h; y++)
          if (fread(p + (size_t)y * stride, 1, w, fin) != (size_t)w) { ok = false; break; }
      }
      if (!ok) { delete img; img = nullptr; }
    }
    if (!img) { en265_push_eof(ectx); eof = true; }
    else en265_push_image(ectx, img);

    en265_encode(ectx);
    for (;;) {
      en265_packet* pck = en265_get_packet(ectx, 0);
      if (!pck) break;
      /* Annex-B start code + NAL, as PacketSink_File does (image-io.cc) */
      const uint8_t sc[3] = {0, 0, 1};
      fwrite(sc, 1, 3, fout);
      fwrite(pck->data, 1, pck->length, fout);
      en265_free_packet(ectx, pck);
    }
  }
  fclose(fout);
  fclose(fin);
  en265_free_encoder(ectx);
  de265_free();
  return 0;
}
