// Cross-check driver (build container only): decodes files produced by the oracle with the
// REFERENCE's independent JavaScript decoder (web/mic-decoder.js, imported from where it lies
// under /root/reference -- never copied into this repo) and writes the decoded samples.
//   node js_crosscheck.mjs <decoder.mjs> <job.json>
// job.json: [{kind:"file"|"frame"|"mic2", in:<path>, out:<path>, width, height, frame}]
import { readFileSync, writeFileSync } from 'fs';
import { pathToFileURL } from 'url';
const [, , decoderPath, jobPath] = process.argv;
import(pathToFileURL(decoderPath).href).then((mod) => {
  const MICDecoder = mod.MICDecoder;
  const jobs = JSON.parse(readFileSync(jobPath, 'utf8'));
  for (const j of jobs) {
    const bytes = new Uint8Array(readFileSync(j.in));
    let out;
    if (j.kind === 'frame') out = MICDecoder.decode(bytes, j.width, j.height);
    else if (j.kind === 'mic2') {
      const hdr = MICDecoder.parseMIC2Header(bytes);
      let prev = null; const parts = [];
      for (let f = 0; f < hdr.frameCount; f++) { prev = MICDecoder.decodeMIC2Frame(bytes, f, prev, hdr); parts.push(prev); }
      out = new Uint16Array(parts.length * parts[0].length);
      parts.forEach((p, i) => out.set(p, i * p.length));
    } else {
      const r = MICDecoder.decodeFile(bytes);
      out = r.pixels ? r.pixels : r.rgb;
    }
    writeFileSync(j.out, Buffer.from(out.buffer, out.byteOffset, out.byteLength));
  }
  console.log('ok ' + jobs.length);
}).catch((e) => { console.error(e.stack || e); process.exit(1); });
