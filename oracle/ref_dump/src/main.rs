//! Writes small indexes with the REAL crate, runs fixed query sets through `Search::search` and dumps the crate's answers:
//! the bytes (index.bin / vector.bin / delete.bin) and the scores this repo's loaders and kernels are to be pinned against
//! (tests/test_ref_dump.py consumes the output when it exists).  Mirrors the calls of the reference's own tests
//! (tests/test.rs:20-210 lexical, 427-470 delete, 614-760 external vectors).  NOT built in this repo's image (no cargo): a recipe for a
//! box that has it; see Cargo.toml.  One shard everywhere, so that shards/0/ holds the whole corpus.
//!
//! Output layout (under the directory given as the first argument, default `dump`):
//!   <case>/index/...         the index directory the crate wrote
//!   <case>/expected.json     {"case", "schema", "meta": {...}, "docs": n, "deleted": [...], "queries": [{query, query_type, result_type,
//!                             mode, k, doc_ids, scores, result_count, result_count_total, (query_vector)}]}
//! Cases:
//!   single      one indexed field, NgramSet::SingleTerm                      (key heads of 20 bytes)
//!   single_del  the same corpus after delete_document on a few docs           (delete.bin)
//!   ngram       one indexed field, NgramFF | NgramFFF + English frequent words (key heads of 22 / 23 bytes, n-gram keys, phrases over them)
//!   fields      title / body / url with boosts 2 / 1 / 0.5                    (BM25F, several length-byte planes per level)
//!   levels      70 000 docs of the single-field schema                        (two 65 536-doc levels in one index.bin)
//!   vector_f32  Inference::External, 64 dims, Precision::F32, Quantization::None, Dot   (vector.bin, f32 records)
//!   vector_i8   the same rows, Quantization::ScalarQuantizationI8                         (vector.bin, i8 records: scale / zero point / sum_q)
use seekstorm::commit::Commit;
use seekstorm::index::{
    AccessType, Close, Clustering, DeleteDocument, DocumentCompression, FrequentwordType, IndexArc, IndexDocuments, IndexMetaObject,
    LexicalSimilarity, NgramSet, StemmerType, StopwordType, TokenizerType, create_index, open_index,
};
use seekstorm::search::{QueryRewriting, QueryType, ResultType, Search, SearchMode};
use seekstorm::vector::{Embedding, Inference, Precision, Quantization};
use seekstorm::vector_similarity::{AnnMode, VectorSimilarity};
use std::{env, fs, path::Path};

/// key_hash of a SingleTerm key as the crate computes it WITHOUT its gxhash path (index.rs:4130-4225: ahash RandomState with fixed seeds,
/// low three bits = NgramType cleared).  hash64 is pub(crate), so the dumper restates it; tests/test_ref_dump.py checks every hash it
/// is given against the file (the key must exist and hold exactly the postings the corpus generator says).  A build of the crate WITH
/// gxhash + AES target features hashes differently: the test then reports the keys as missing instead of comparing wrong lists.
fn key_hash(term: &str) -> u64 {
    use std::hash::BuildHasher;
    ahash::RandomState::with_seeds(808259318, 750368348, 84901999, 789810389).hash_one(term.as_bytes()) & !7u64
}
fn term_keys(queries: &[(&str, QueryType)]) -> serde_json::Value {
    let mut m = serde_json::Map::new();
    for (q, _) in queries {
        for tok in q.split(|c: char| !c.is_alphanumeric()).filter(|t| !t.is_empty()) {
            m.insert(tok.to_string(), serde_json::json!(key_hash(tok).to_string()));
        }
    }
    serde_json::Value::Object(m)
}

fn meta(name: &str, ngram: u8, frequent: FrequentwordType, inference: Inference) -> IndexMetaObject {
    IndexMetaObject {
        id: 0,
        name: name.into(),
        lexical_similarity: LexicalSimilarity::Bm25f,
        tokenizer: TokenizerType::UnicodeAlphanumeric,
        stemmer: StemmerType::None,
        stop_words: StopwordType::None,
        frequent_words: frequent,
        ngram_indexing: ngram,
        document_compression: DocumentCompression::Snappy,
        access_type: AccessType::Mmap,
        spelling_correction: None,
        query_completion: None,
        clustering: Clustering::None,
        inference,
    }
}

/// doc i holds word w<j> for the j with i % (j + 2) == 0, repeated (i / 7) % (j + 1) + 1 times, then two pad words; every third doc
/// carries the fixed sequence "the quick w0 w1 of the day" (frequent words around the planted phrase: n-gram keys in the ngram case)
fn body(i: u32) -> String {
    let mut s = String::new();
    if i % 3 == 0 {
        s.push_str("the quick w0 w1 of the day ");
    }
    for j in 0..12u32 {
        if i % (j + 2) == 0 {
            for _ in 0..(1 + (i / 7) % (j + 1)) {
                s.push_str(&format!("w{} ", j));
            }
        }
    }
    s.push_str(&format!("pad{} pad{}", i % 97, i % 89));
    s
}

fn lexical_queries() -> Vec<(&'static str, QueryType)> {
    vec![
        ("w0 w1", QueryType::Union),
        ("w0 w1 w2", QueryType::Union),
        ("w2 w5 w7 w9 w11", QueryType::Union),
        ("+w3 +w5", QueryType::Intersection),
        ("+w0 +w1 +w2", QueryType::Intersection),
        ("w7", QueryType::Union),
        ("\"w0 w1\"", QueryType::Phrase),
        ("\"the quick w0\"", QueryType::Phrase),
        ("\"of the day\"", QueryType::Phrase),
        ("w2 -w3", QueryType::Union),
        ("+w2 +w4 -w5", QueryType::Intersection),
        ("w0 w1 w2 w3 w4 w5 w6 w7 w8 w9 w10 w11", QueryType::Union), // 12 terms: union_blockid -> union_scan_32
    ]
}

async fn run_lexical(index_arc: &IndexArc, k: usize) -> Vec<serde_json::Value> {
    let mut out = Vec::new();
    for (q, qt) in lexical_queries() {
        for rt in [ResultType::Topk, ResultType::TopkCount, ResultType::Count] {
            let ro = index_arc
                .search(q.to_string(), None, qt.clone(), SearchMode::Lexical, false, 0, k, rt.clone(), false, Vec::new(), Vec::new(), Vec::new(),
                        Vec::new(), QueryRewriting::SearchOnly)
                .await;
            out.push(serde_json::json!({
                "query": q, "query_type": format!("{:?}", qt), "result_type": format!("{:?}", rt), "mode": "Lexical", "k": k,
                "doc_ids": ro.results.iter().map(|r| r.doc_id).collect::<Vec<_>>(),
                "scores": ro.results.iter().map(|r| r.score).collect::<Vec<_>>(),
                "result_count": ro.result_count, "result_count_total": ro.result_count_total,
            }));
        }
    }
    out
}

async fn lexical_case(out: &Path, case: &str, schema_json: &str, ngram: u8, frequent: FrequentwordType, n_docs: u32, fields: bool,
                      delete: &[u64]) {
    let dir = out.join(case);
    let index_path = dir.join("index");
    let _ = fs::remove_dir_all(&dir);
    fs::create_dir_all(&index_path).unwrap();
    let schema = serde_json::from_str(schema_json).unwrap();
    let index_arc = create_index(&index_path, meta(case, ngram, frequent, Inference::None), &schema, &Vec::new(), 11, false, Some(1)).await.unwrap();
    let mut docs = Vec::new();
    for i in 0..n_docs {
        if fields {
            docs.push(serde_json::json!({ "title": format!("w{} w{} title{}", i % 5, i % 7, i % 13), "body": body(i), "url": format!("site{} w{}", i % 31, i % 3) }));
        } else {
            docs.push(serde_json::json!({ "body": body(i) }));
        }
    }
    index_arc.index_documents(serde_json::from_value(serde_json::Value::Array(docs)).unwrap()).await;
    index_arc.commit().await;
    for d in delete {
        index_arc.delete_document(*d).await;
    }
    index_arc.close().await;
    let index_arc = open_index(&index_path).await.unwrap();
    let mut queries = run_lexical(&index_arc, 10).await;
    queries.extend(run_lexical(&index_arc, 100).await);
    let expected = serde_json::json!({ "case": case, "schema": serde_json::from_str::<serde_json::Value>(schema_json).unwrap(),
        "meta": { "ngram_indexing": ngram, "shards": 1 }, "docs": n_docs, "deleted": delete, "term_keys": term_keys(&lexical_queries()),
        "queries": queries });
    fs::write(dir.join("expected.json"), serde_json::to_string_pretty(&expected).unwrap()).unwrap();
    index_arc.close().await;
}

/// row r, component c: a fixed pseudo-random value in [-1, 1) (an LCG on (r, c)); rows are NOT normalised by the dumper -- the crate
/// applies what its similarity asks for
fn vec_row(r: u32, dim: usize) -> Vec<f32> {
    let mut v = Vec::with_capacity(dim);
    for c in 0..dim as u32 {
        let x = (r.wrapping_mul(2654435761).wrapping_add(c.wrapping_mul(40503)).wrapping_mul(1103515245).wrapping_add(12345) >> 8) & 0xFFFF;
        v.push(x as f32 / 32768.0 - 1.0);
    }
    v
}

async fn vector_case(out: &Path, case: &str, quantization: Quantization, n_rows: u32, dim: usize) {
    let dir = out.join(case);
    let index_path = dir.join("index");
    let _ = fs::remove_dir_all(&dir);
    fs::create_dir_all(&index_path).unwrap();
    let schema_json = r#"[{"field":"vector","field_type":"Json","store":false,"index_lexical":false,"index_vector":true},
                          {"field":"index","field_type":"Text","store":true,"index_lexical":false,"index_vector":false}]"#;
    let schema = serde_json::from_str(schema_json).unwrap();
    let inference = Inference::External { dimensions: dim, precision: Precision::F32, quantization, similarity: VectorSimilarity::Dot };
    let index_arc = create_index(&index_path, meta(case, NgramSet::SingleTerm as u8, FrequentwordType::None, inference), &schema, &Vec::new(), 11,
                                 false, Some(1)).await.unwrap();
    let mut docs = Vec::new();
    for r in 0..n_rows {
        docs.push(serde_json::json!({ "vector": vec_row(r, dim), "index": format!("{}", r) }));
    }
    index_arc.index_documents(serde_json::from_value(serde_json::Value::Array(docs)).unwrap()).await;
    index_arc.commit().await;
    index_arc.close().await;
    let index_arc = open_index(&index_path).await.unwrap();
    let mut queries = Vec::new();
    for qi in 0..8u32 {
        let qv = vec_row(1_000_000 + qi, dim);
        for k in [10usize, 100] {
            let ro = index_arc
                .search(String::new(), Some(Embedding::F32(qv.clone())), QueryType::Union,
                        SearchMode::Vector { similarity_threshold: None, ann_mode: AnnMode::All }, false, 0, k, ResultType::TopkCount, false,
                        Vec::new(), Vec::new(), Vec::new(), Vec::new(), QueryRewriting::SearchOnly)
                .await;
            queries.push(serde_json::json!({
                "query": "", "query_type": "Union", "result_type": "TopkCount", "mode": "Vector", "k": k, "query_vector": qv,
                "doc_ids": ro.results.iter().map(|r| r.doc_id).collect::<Vec<_>>(),
                "scores": ro.results.iter().map(|r| r.score).collect::<Vec<_>>(),
                "result_count": ro.result_count, "result_count_total": ro.result_count_total,
            }));
        }
    }
    let expected = serde_json::json!({ "case": case, "schema": serde_json::from_str::<serde_json::Value>(schema_json).unwrap(),
        "meta": { "dimensions": dim, "similarity": "Dot", "shards": 1 }, "docs": n_rows, "deleted": [], "queries": queries });
    fs::write(dir.join("expected.json"), serde_json::to_string_pretty(&expected).unwrap()).unwrap();
    index_arc.close().await;
}

#[tokio::main]
async fn main() {
    let out = env::args().nth(1).unwrap_or_else(|| "dump".into());
    let out = Path::new(&out);
    fs::create_dir_all(out).unwrap();
    let one = r#"[{"field":"body","field_type":"Text","store":true,"index_lexical":true,"longest":true}]"#;
    let three = r#"[{"field":"title","field_type":"Text","store":true,"index_lexical":true,"boost":2.0},
                    {"field":"body","field_type":"Text","store":true,"index_lexical":true,"longest":true},
                    {"field":"url","field_type":"Text","store":true,"index_lexical":true,"boost":0.5}]"#;
    let single = NgramSet::SingleTerm as u8;
    lexical_case(out, "single", one, single, FrequentwordType::None, 5000, false, &[]).await;
    lexical_case(out, "single_del", one, single, FrequentwordType::None, 5000, false, &[0, 6, 12, 30, 2310, 4620]).await;
    lexical_case(out, "ngram", one, NgramSet::NgramFF as u8 | NgramSet::NgramFFF as u8, FrequentwordType::English, 5000, false, &[]).await;
    lexical_case(out, "fields", three, single, FrequentwordType::None, 5000, true, &[]).await;
    lexical_case(out, "levels", one, single, FrequentwordType::None, 70000, false, &[]).await;
    vector_case(out, "vector_f32", Quantization::None, 3000, 64).await;
    vector_case(out, "vector_i8", Quantization::ScalarQuantizationI8, 3000, 64).await;
    println!("wrote {}", out.display());
}
