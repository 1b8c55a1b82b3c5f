/*
 * aac.js_amd/js/main.js — the package's `main`, shaped like the reference's (package.json:5 -> src/decoder.js): requiring it
 * registers the decoder for 'mp4a' and 'aac ' and the ADTS demuxer with Aurora as a side effect (src/decoder.js:29-31,
 * src/adts_demuxer.js:4-5) and exports the decoder class (src/decoder.js:437).
 *
 *   require('av'); require('aac.js_amd');         // instead of require('aac')
 *
 * Options (look-ahead, TNS / PNS / coupling modes, int16 PCM, the device front end) come from the environment variable
 * AACG_OPTIONS as JSON, e.g. {"lookahead":64,"gpuParse":true}; an application that wants them in code calls
 * require('aac.js_amd/js/aurora.js').register(AV, options) itself.
 */
'use strict';
const AV = require('av');
let options = {};
if (process.env.AACG_OPTIONS) {
    try { options = JSON.parse(process.env.AACG_OPTIONS); } catch (e) { throw new Error('AACG_OPTIONS is not JSON: ' + e.message); }
}
const plugin = require('./aurora.js').register(AV, options);
module.exports = plugin.Decoder;
module.exports.Demuxer = plugin.Demuxer;
